#!/usr/bin/env python3
"""HBM-side counter traffic of the kernels OTHER than the headline's, per launch, against their algorithmic bytes:
tools/pmc_hbm_others.sh TAG wrote profiles/<TAG>_pmc_<workload>_{FETCH,WRITE}_SIZE.txt (separate rocprofv3 --pmc passes of
tools/traffic_workload.py with TRAFFIC_WORKLOAD = c2 | c4v | c5 | c5r4 | fdn16 | rv3); corrections as tools/pmc_latest.py (FETCH_SIZE x2 on gfx950, both
in KB), checked on the known 12.58 GB copy of the same pass.
usage: tools/pmc_others_view.py TAG [TAG ...]"""
import os
import re
import sys

T = 48000
ALGO = {"c5": (2048 * T * 272, "k_fdn_render_frames", 1.0, "2 048 x reverb_stereo(10, 2, 0.5) x 48 000 frames, 272 B per instance-frame"),
        "c5r4": (2048 * T * 272, "k_fdn_render_frames", 1.0, "2 048 x reverb4_stereo(20, 2) x 48 000 frames, 272 B per instance-frame"),
        "fdn16": (4096 * T * 136, "k_fdn_frames_generic", 1.0, "4 096 x the prelude's fdn example (16 lines) x 48 000 frames, 136 B per instance-frame"),
        "rv3": (2048 * T * 624, "k_rv3_render", 1.0, "2 048 x reverb3_stereo(2, 0.5, lowpole_hz(8000)) x 48 000 frames, 624 B per instance-frame"),
        "c2": (65536 * T * 4, "k_render", 1.0, "65 536 x noise >> biquad x 48 000 frames, 4 B per voice-sample"),
        # 7 dispatches = the 64-frame priming block + 3 notes of two 24 000-frame launches: the mean is over 7, the bytes over 6
        "c4v": (32768 * (T // 2) * 8, "k_render_pipe", 7.0 / 6.0, "32 768 x config-4 voice (Var gate) x 24 000 frames per launch, 8 B per voice-sample")}


def means(path, counter):
    out, kernel = {}, None
    for line in open(path):
        if line.startswith("kernel:"):
            kernel = line.split("kernel:")[1].strip()
        m = re.match(r"\s+" + counter + r"\s+dispatches=\s*(\d+) mean=([0-9.e+]+)", line)
        if m and kernel:
            out[kernel] = (float(m.group(2)), int(m.group(1)))
    return out


for tag in sys.argv[1:]:
    for wl, (algo, pick, fix, what) in ALGO.items():
        pf, pw = f"profiles/{tag}_pmc_{wl}_FETCH_SIZE.txt", f"profiles/{tag}_pmc_{wl}_WRITE_SIZE.txt"
        if not (os.path.exists(pf) and os.path.exists(pw)):
            continue
        f, w = means(pf, "FETCH_SIZE"), means(pw, "WRITE_SIZE")
        ck = [k for k in f if "copyBuffer" in k][0]
        known = 65536 * 48000 * 4 // f[ck][1]
        fs, ws = known / (f[ck][0] * 1024), known / (w[ck][0] * 1024)
        for k in f:
            if pick in k and "lifecycle" not in k:
                fb, wb = f[k][0] * 1024 * round(fs) * fix, w[k][0] * 1024 * round(ws) * fix
                print(f"{tag} {wl:5s} {what}\n      kernel {k[:70]}\n      fetched {fb / 1e9:7.3f} GB  written {wb / 1e9:7.3f} GB  total {(fb + wb) / 1e9:7.3f} GB"
                      f"  algorithmic {algo / 1e9:7.3f} GB  ratio {(fb + wb) / algo:.4f}   (copy calibration: FETCH x{fs:.3f}, WRITE x{ws:.3f})")
