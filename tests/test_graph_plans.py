"""Host logic of Bank.from_graph's recognisers (no GPU): which graphs are a gain / dry bus around a node with a lane-per-frame kernel
(graph.bus_plan -> fdsp_bank_set_bus), in every spelling the reference's documentation uses (README.md:436, wave.rs:514, CHANGES.md:203,
net.rs:681), and which are not."""
import numpy as np

from fundsp_amd import BUS_DRY_WET, BUS_WET
from fundsp_amd import graph as G


def rv():
    return G.reverb_stereo(20.0, 2.0, 1.0)


def test_bus_plan_recognises_the_documented_spellings():
    f = np.float32
    cases = [
        (G.multipass(2) & 0.2 * rv(), (BUS_DRY_WET, f(0.2), f(1.0))),                  # README.md:436
        (0.2 * rv() & G.multipass(2), (BUS_DRY_WET, f(0.2), f(1.0))),                  # wave.rs:514
        (0.3 * rv() & (1.0 - 0.3) * G.multipass(2), (BUS_DRY_WET, f(0.3), f(0.7))),    # CHANGES.md:203
        (G.multipass(2) & rv(), (BUS_DRY_WET, f(1.0), f(1.0))),                        # net.rs:681
        (rv() * 0.5, (BUS_WET, f(0.5), f(1.0))),
    ]
    for g, want in cases:
        inner, mode, wet, dry = G.bus_plan(g)
        assert (mode, f(wet), f(dry)) == want, (g.type, mode, wet, dry)
        assert G.lane_per_frame_shape(inner) and inner.stock_reverb[0] == "reverb_stereo"
        assert (g.nin, g.nout) == (2, 2)


def test_bus_plan_covers_every_lane_per_frame_node():
    d = [0.011 + 0.001 * i for i in range(8)]
    mono = G.split(8) >> G.fdn(G.stacki(8, lambda i: G.delay(d[i]) >> G.fir(0.5, 0.4))) >> G.join(8)
    for node, p in ((G.reverb4_stereo(10.0, 3.0), G.multipass(2)), (G.reverb3_stereo(2.0, 0.6, lambda: G.lowpole_hz(6000.0)), G.multipass(2)), (mono, G.pass_())):
        plan = G.bus_plan(p & 0.25 * node)
        assert plan is not None and plan[1] == BUS_DRY_WET and G.lane_per_frame_shape(plan[0])


def test_bus_plan_leaves_other_shapes_alone():
    assert G.bus_plan(rv()) is None                                    # the bare node: nothing to fold
    assert G.bus_plan(rv() & rv()) is None                             # a Bus of two reverbs: no pass side
    assert G.bus_plan(G.multipass(2) & np.array([0.1, 0.2], dtype=np.float32) * rv()) is not None   # (the wet side is found ...)
    assert G.bus_plan(np.array([0.1, 0.2], dtype=np.float32) * rv()) is None                        # ... but a per-instance factor is no uniform bus
    inner, mode, wet, dry = G.bus_plan(G.multipass(2) & np.array([0.1, 0.2], dtype=np.float32) * rv())
    assert not G.lane_per_frame_shape(inner)                           # its inner graph is the scaled reverb, which has no lane-per-frame kernel: compiled whole
    plan = G.bus_plan(0.5 * G.noise())                                 # a gain on something else is a plan, but not around a lane-per-frame node
    assert plan is not None and not G.lane_per_frame_shape(plan[0])
    assert G.bus_plan((G.noise() | G.noise()) >> (G.multipass(2) & 0.2 * rv())) is None   # the Pipe itself is no bus (its right side is: Bank.from_graph's chain)
