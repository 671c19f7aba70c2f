"""The oracle's combinators against linear-systems truth -- CPU only.

The reference checks its graphs against analytic frequency responses (tests/test_flow.rs:18-80 `test_response`, cases
:85-177 with bus / branch / split / join / stacki / pipei / reverse networks).  Its response algebra (signal.rs) is not
part of the path, but the same truth can be had from the oracle itself: for LINEAR nodes the transfer matrix of a
combinator is a fixed function of its operands' transfer matrices (Pipe = product, Bus = sum, Branch = rows stacked,
Stack = block diagonal, `+` / `-` = columns side by side, Thru = identity below the node's rows, ...).  Random trees over
linear leaves -- whose own responses test_oracle_response.py pins to the closed forms -- are measured as a whole and
compared with the composition of their parts."""
import numpy as np
import pytest

import oracle as O

N = 4096


def leaf(rng, nin, nout):
    r = lambda lo, hi: float(np.float32(rng.uniform(lo, hi)))
    pool = {
        (1, 1): [lambda: ("lowpass_hz", r(300, 8000), r(0.5, 2)), lambda: ("highpole_hz", r(50, 2000)),
                 lambda: ("lowpole_hz", r(300, 9000)), lambda: ("pass_",), lambda: ("mul", r(-1.5, 1.5)), lambda: ("tick",),
                 lambda: ("bell_hz", r(300, 5000), r(0.5, 2), r(0.5, 2)), lambda: ("delay", r(0.0001, 0.002)),
                 lambda: ("fir", r(-1, 1), r(-1, 1), r(-1, 1)), lambda: ("allpole_delay", r(0.1, 1.5)),
                 lambda: ("butterpass_hz", r(300, 6000)), lambda: ("resonator_hz", r(300, 5000), r(100, 800))],
        (1, 2): [lambda: ("pan", r(-1, 1)), lambda: ("split", 2)],
        (2, 1): [lambda: ("join", 2)],
        (2, 2): [lambda: ("reverse", 2), lambda: ("multipass", 2), lambda: ("rotate", r(0, 3), r(0.3, 1.0))],
        (1, 0): [lambda: ("sink",)],
        (2, 0): [lambda: ("multisink", 2)],
    }
    opts = pool.get((nin, nout))
    return None if not opts else opts[rng.integers(len(opts))]()


def gen(rng, nin, nout, depth):
    choices = []
    if depth > 0:
        choices += ["pipe"] * 3
        if nin >= 2 or (nin >= 1 and nout >= 2):
            choices.append("stack")
        if nout >= 1:
            choices += ["bus", "unop"]
            if nin >= 2:
                choices.append("binop")
        if nout == 2:
            choices.append("branch")
        if nin == nout:
            choices.append("thru")
    lf = leaf(rng, nin, nout)
    if lf is not None:
        choices += ["leaf"] * (2 if depth > 0 else 50)
    for _ in range(100):
        c = choices[rng.integers(len(choices))]
        if c == "leaf":
            return lf
        if c == "pipe":
            k = int(rng.integers(1, 3))
            return ("pipe", gen(rng, nin, k, depth - 1), gen(rng, k, nout, depth - 1))
        if c == "stack":  # every operand keeps at least one input: there are no linear generators
            a, b = int(rng.integers(1, nin)) if nin >= 2 else 0, int(rng.integers(0, nout + 1))
            if a == 0 or nin - a == 0:
                continue
            return ("stack", gen(rng, a, b, depth - 1), gen(rng, nin - a, nout - b, depth - 1))
        if c == "bus":
            return ("bus", gen(rng, nin, nout, depth - 1), gen(rng, nin, nout, depth - 1))
        if c == "branch":
            return ("branch", gen(rng, nin, 1, depth - 1), gen(rng, nin, 1, depth - 1))
        if c == "thru":
            return ("thru", gen(rng, nin, int(rng.integers(0, nin + 1)), depth - 1))
        if c == "binop":
            a = int(rng.integers(1, nin))
            return ("binop", "+-"[rng.integers(2)], gen(rng, a, nout, depth - 1), gen(rng, nin - a, nout, depth - 1))
        if c == "unop":
            return ("unop", ["mul", "neg"][rng.integers(2)], float(np.float32(rng.uniform(-1.5, 1.5))), gen(rng, nin, nout, depth - 1))
    raise AssertionError("no production fits")


def build(t):
    k = t[0]
    if k == "pipe": return build(t[1]) >> build(t[2])
    if k == "stack": return build(t[1]) | build(t[2])
    if k == "bus": return build(t[1]) & build(t[2])
    if k == "branch": return build(t[1]) ^ build(t[2])
    if k == "thru": return ~build(t[1])
    if k == "binop": return build(t[2]) + build(t[3]) if t[1] == "+" else build(t[2]) - build(t[3])
    if k == "unop": return build(t[3]) * t[2] if t[1] == "mul" else -build(t[3])
    return getattr(O, k)(*t[1:])


def measure(node):
    """Transfer matrix [outputs][inputs][bins] from one impulse per input (fresh state each time via reset)."""
    nin, nout = node.inputs(), node.outputs()
    H = np.zeros((nout, nin, N // 2 + 1), dtype=np.complex128)
    for i in range(nin):
        node.reset()
        x = np.zeros((nin, N), dtype=np.float32)
        x[i, 0] = 1.0
        y = node.render_blocks(x)
        for o in range(nout):
            H[o, i] = np.fft.rfft(y[o].astype(np.float64))
    return H


def expected(t):
    k = t[0]
    if k == "pipe":
        a, b = expected(t[1]), expected(t[2])
        return np.einsum("omk,mik->oik", b, a)
    if k == "stack":
        a, b = expected(t[1]), expected(t[2])
        H = np.zeros((a.shape[0] + b.shape[0], a.shape[1] + b.shape[1], a.shape[2]), dtype=np.complex128)
        H[:a.shape[0], :a.shape[1]] = a
        H[a.shape[0]:, a.shape[1]:] = b
        return H
    if k == "bus":
        return expected(t[1]) + expected(t[2])
    if k == "branch":
        return np.concatenate([expected(t[1]), expected(t[2])], axis=0)
    if k == "thru":                       # x's outputs, then the inputs x has no output for (cut if x has more)
        a = expected(t[1])
        nin = a.shape[1]
        H = np.zeros((nin, nin, a.shape[2]), dtype=np.complex128)
        for o in range(nin):
            if o < a.shape[0]:
                H[o] = a[o]
            else:
                H[o, o] = 1.0
        return H
    if k == "binop":                      # inputs side by side, outputs added / subtracted
        a, b = expected(t[2]), expected(t[3])
        return np.concatenate([a, b if t[1] == "+" else -b], axis=1)
    if k == "unop":
        a = expected(t[3])
        return a * t[2] if t[1] == "mul" else -a
    return measure(build(t))              # a leaf: its own measured response


@pytest.mark.parametrize("seed", range(40))
def test_combinators_compose_like_linear_systems(seed):
    rng = np.random.default_rng(500 + seed)
    nin, nout = int(rng.integers(1, 3)), int(rng.integers(1, 3))
    tree = gen(rng, nin, nout, depth=int(rng.integers(2, 5)))
    node = build(tree)
    assert (node.inputs(), node.outputs()) == (nin, nout), tree
    got, want = measure(node), expected(tree)
    scale = max(1.0, float(np.abs(want).max()))
    assert np.abs(got - want).max() <= 2e-4 * scale, (tree, float(np.abs(got - want).max()), scale)
