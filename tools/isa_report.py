#!/usr/bin/env python3
"""Static ISA report of the gfx950 kernels in one translation unit: registers, occupancy, LDS, scratch and instruction mix.

    python tools/isa_report.py cost_volume_fwd.hip [--match nhwc_kernelILi8] [--dump /tmp/k.s]

Compiles with the library's flags (deep_reconstruction_with_epipolar_lines_mvster_amd/build.py) to assembly; no GPU needed.
The loop bodies of the cost-volume kernels are not unrolled over the source views, so "static VALU" is roughly
setup + one view."""
import argparse
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from deep_reconstruction_with_epipolar_lines_mvster_amd import build as B    # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("source")
    ap.add_argument("--match", default="")
    ap.add_argument("--dump", default=None, help="write the assembly of the first matching kernel here")
    ap.add_argument("-D", action="append", default=[])
    args = ap.parse_args()
    src = args.source if os.path.exists(args.source) else os.path.join(B.CSRC, args.source)
    with tempfile.TemporaryDirectory() as tmp:
        out = os.path.join(tmp, "k.s")
        cmd = [B._hipcc()] + B.FLAGS + ["-I", B.INCLUDE, "-I", B.CSRC, "-S", "--cuda-device-only", src, "-o", out] + ["-D" + d for d in args.D]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            sys.exit(r.stderr)
        text = open(out).read()
    # split per kernel: "<name>:" ... "s_endpgm" ... metadata comments up to the next symbol
    starts = [(m.start(), m.group(1)) for m in re.finditer(r"^(_Z\w+):\s*; @", text, re.M)]
    dumped = False
    for i, (pos, name) in enumerate(starts):
        end = starts[i + 1][0] if i + 1 < len(starts) else len(text)
        body = text[pos:end]
        dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
        short = re.sub(r"\(.*", "", dem).replace("void mvster::", "")
        if args.match and args.match not in name and args.match not in short:
            continue
        code = body[:body.find("s_endpgm")]
        ops = re.findall(r"^\s+([a-z_0-9]+)", code, re.M)
        n = lambda pre: sum(1 for o in ops if o.startswith(pre))       # noqa: E731
        get = lambda key: (re.search(key + r":?\s+(\d+)", body) or [None, "?"])[1]      # noqa: E731
        print("%-58s vgpr %3s sgpr %3s occ %s lds %6s scratch %s | VALU %4d (pk %3d, dpp %2d, trans %2d) SALU %4d vmem %3d lds %3d" % (
            short[:58], get(r"; NumVgprs"), get(r"; NumSgprs"), get(r"; Occupancy"), get(r"; LDSByteSize"), get(r"; ScratchSize"),
            n("v_"), n("v_pk_"), sum(1 for line in code.splitlines() if "dpp" in line or "quad_perm" in line),
            sum(1 for o in ops if o in ("v_rcp_f32_e32", "v_exp_f32_e32", "v_rsq_f32_e32", "v_sqrt_f32_e32", "v_log_f32_e32")),
            n("s_"), n("global_") + n("buffer_") + n("flat_"), n("ds_")))
        if args.dump and not dumped:
            open(args.dump, "w").write(body)
            dumped = True


if __name__ == "__main__":
    main()
