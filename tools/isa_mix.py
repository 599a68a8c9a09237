"""Instruction mix of the MFMA-carrying basic blocks of the hot kernels, from hipcc's own assembly listing.

    python tools/isa_mix.py [file.hip ...]        (default: the K-loop kernels of gan_ffn_amd/csrc)

For every kernel of a translation unit and every basic block with at least four MFMAs: instructions, MFMAs, accumulator moves
(v_accvgpr_read / write / mov), other vector-ALU instructions, LDS and global / buffer memory instructions.  fp32 MFMAs execute on
the SIMD's vector ALU, so the "valu" and "accmov" columns are paid in MFMA time: this listing is how round 5
found the accumulator shuffling of tn100_kernel, the per-lane 64-bit addresses in every K loop and the select chains of the
K = 100 epilogues.  Compiles with the flags of csrc/Makefile; no GPU needed."""
import os, re, subprocess, sys, tempfile
from collections import Counter

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "gan_ffn_amd", "csrc")
DEFAULT = ["gemm.hip", "gemm_n100.hip", "gemm_tn100.hip"]


def demangle(names):
    try:
        out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.split("\n")
        return dict(zip(names, out))
    except Exception:
        return {n: n for n in names}


def listing(path):
    with tempfile.TemporaryDirectory() as d:
        out = os.path.join(d, "k.s")
        subprocess.run(["hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950:xnack-", "-mllvm", "-amdgpu-mfma-vgpr-form=1", "--cuda-device-only", "-S", "-o", out, os.path.basename(path)],
                       cwd=os.path.dirname(path), check=True, stderr=subprocess.DEVNULL)
        return open(out).read()


def blocks_of(text):
    for m in re.finditer(r"^(_Z\w+):\s*; @", text, re.M):
        name, i = m.group(1), m.end()
        body = text[i:text.index(".Lfunc_end", i)].split("\n")
        lab, cur, rows = "entry", [], []
        for line in body + [".LBB0_0:"]:
            if re.match(r"^\.LBB\d+_\d+:", line):
                ins = [x.strip().split()[0] for x in cur if x.strip() and not x.strip().startswith((";", "."))]
                mf = sum("mfma" in x for x in ins)
                if mf >= 4:
                    c = Counter()
                    for x in ins:
                        if "mfma" in x: c["mfma"] += 1
                        elif x.startswith("v_accvgpr"): c["accmov"] += 1
                        elif x.startswith("v_"): c["valu"] += 1
                        elif x.startswith("ds_"): c["lds"] += 1
                        elif x.startswith(("global_", "buffer_", "flat_")): c["vmem"] += 1
                    rows.append((lab, len(ins), c))
                lab, cur = line.split(":")[0], []
            else:
                cur.append(line)
        if rows:
            yield name, rows


def main(files):
    for f in files:
        path = f if os.path.isabs(f) else os.path.join(CSRC, f)
        text = listing(path)
        found = list(blocks_of(text))
        names = demangle([n for n, _ in found])
        print("== %s" % os.path.basename(path))
        for name, rows in found:
            print(names.get(name, name).replace("ganffn::", "").replace("(anonymous namespace)::", "")[:118])
            for lab, n, c in rows:
                print("   %-10s ins %4d  mfma %3d  accmov %3d  valu %3d  lds %3d  vmem %3d" % (lab, n, c["mfma"], c["accmov"], c["valu"], c["lds"], c["vmem"]))


if __name__ == "__main__":
    main(sys.argv[1:] or DEFAULT)
