"""Memory-wait skeleton of one kernel: compiles a .hip file of neuralmonkey_amd/csrc to gfx950 assembly (device only,
no GPU needed) and prints, in program order, the loads, stores, waits, barriers, LDS / DPP cross-lane operations and
the first MFMA / transcendental of the kernel whose mangled name contains the given substring.

    python tools/isa_waits.py nm_attention.hip attn_whole_fastILi13E
    python tools/isa_waits.py nm_gemm.hip gemm_skinny16ILi16ELb0E --rev <commit>      # the file as of a commit

What to look for (MI355X_MICROARCH.md: vmcnt counts loads in ISSUE order):
  * `s_waitcnt vmcnt(0)` in front of the first arithmetic: something the first instruction needs was requested after
    everything else, so nothing starts before the last load has landed;
  * `ds_bpermute_b32` chains with `s_waitcnt lgkmcnt(0)` in between: __shfl_xor reductions -- one LDS-pipe round trip
    per step (use the DPP steps of nm_wave_sum_dpp where all 64 lanes are active);
  * waits directly behind prefetched operands: values that meet in phi registers (one load per branch) are waited for
    where the branches join."""
import argparse
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KEEP = ("global_load", "global_store", "buffer_load", "buffer_store", "s_waitcnt", "s_barrier", "ds_bpermute", "ds_read",
        "ds_write", "s_load", "global_atomic", "v_readlane")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("source")
    ap.add_argument("kernel")
    ap.add_argument("--rev", default=None, help="git revision of the source file (default: the working tree)")
    ap.add_argument("--limit", type=int, default=120)
    args = ap.parse_args()
    src = os.path.join(ROOT, "neuralmonkey_amd", "csrc", args.source)
    with tempfile.TemporaryDirectory() as tmp:
        if args.rev:
            for name in os.listdir(os.path.dirname(src)):
                if name.endswith((".h", ".hip")):
                    blob = subprocess.run(["git", "-C", ROOT, "show", "{}:neuralmonkey_amd/csrc/{}".format(args.rev, name)],
                                          capture_output=True)
                    if blob.returncode == 0:
                        with open(os.path.join(tmp, name), "wb") as fh:
                            fh.write(blob.stdout)
            src = os.path.join(tmp, args.source)
        out = os.path.join(tmp, "k.s")
        hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
        res = subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-S", "--cuda-device-only", src,
                              "-o", out], capture_output=True, text=True)
        if res.returncode != 0:
            sys.exit(res.stderr[-2000:])
        text = open(out).read()
    names = [n for n in re.findall(r"^(_Z\w+):", text, re.M) if args.kernel in n]
    if not names:
        sys.exit("no kernel matches '{}'".format(args.kernel))
    for name in names:
        body = text[text.index(name + ":"):]
        body = body[:body.index(".Lfunc_end")]
        ops = [ln.strip() for ln in body.split("\n") if ln.strip() and not ln.strip().startswith((".", ";"))][1:]
        print("{}  ({} instructions)".format(name, len(ops)))
        shown, first = 0, {"v_mfma": False, "v_exp": False}
        for idx, op in enumerate(ops):
            mnem = op.split()[0]
            hit = mnem.startswith(KEEP) or "_dpp" in mnem or " row_" in op or " quad_perm" in op
            for key in first:
                if mnem.startswith(key) and not first[key]:
                    first[key] = hit = True
            if hit:
                print("{:6d}  {}".format(idx, op[:100]))
                shown += 1
                if shown >= args.limit:
                    print("        ...")
                    break


if __name__ == "__main__":
    main()
