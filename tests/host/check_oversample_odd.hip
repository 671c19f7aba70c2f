// tests/host/check_oversample_odd.hip -- host-side check (hipcc, host only): the Oversampler's process walk (fd_nodes.hpp) over launch blocks of ODD sizes
// against the oracle's Oversampler::process restatement (oversample.rs:178-212).  Per pass the reference hands its inner node a block of `size` samples,
// one more than the 2 (size / 2) it interpolated when `size` is odd -- the zero in the last slot of a zero-initialised buffer; the inner node advances by it.
// Found by the wider-pool fuzzer (tests/test_gpu_graph_fuzz.py) in the ragged last block of a 275-frame launch: the engine skipped that sample, and the oracle
// read it from an uninitialised array.  State is carried from block to block, so a later even block only matches if every odd one before it did.
#include <cstdio>
#include <cstring>
#include <cmath>
#include <vector>
#define FD_HOST_ONLY 1
#include "fd_nodes.hpp"
extern "C" {
#include "fundsp_oracle.h"
}
using namespace fd;
template <class G> void walk(G& g, int size, const float* in, float* out) {   // one launch block the way render_body / wide_fold walk it
    g.begin_block(size);
    const int full = size & ~7;
    for (int f = 0; f < size; f++) {
        float fi[1] = {in[f]}, fo[1];
        if (f < full) g.template step<PH_SIMD>(fi, fo);
        else { if (f == full) g.end_simd(); g.template step<PH_REM>(fi, fo); }
        out[f] = fo[0];
    }
    if (full == size) g.end_simd();
}
int main() {
    int bad = 0;
    const int sizes[] = {64, 19, 1, 7, 64, 63, 9, 1, 1, 64, 33};
    using G = Oversampler<FixedSvf>;
    G g; g.init(); g.x.cutoff = 1500.0f; g.x.q = 0.9f; g.x.gain = 1.0f; g.update(48000.0); g.reset();
    onode* n = o_oversample(o_fixed_svf(O_SVF_LOWPASS, 1500.0f, 0.9f, 1.0f));
    o_set_sample_rate(n, 48000.0);
    unsigned s = 777;
    for (int size : sizes) {
        float x[64], a[64], b[64];
        for (int i = 0; i < 64; i++) { s = s * 1664525u + 1013904223u; x[i] = (float)(s >> 8) * (1.0f / 8388608.0f) - 1.0f; }
        memset(a, 0, sizeof a); memset(b, 0, sizeof b);
        walk(g, size, x, a);
        o_process(n, size, x, b);
        int d = 0;
        for (int i = 0; i < size - (size & 1); i++) d += memcmp(&a[i], &b[i], 4) != 0;
        printf("size %2d: %d samples differ\n", size, d);
        bad += d;
    }
    printf("%s\n", bad ? "MISMATCH" : "all equal");
    return bad != 0;
}
