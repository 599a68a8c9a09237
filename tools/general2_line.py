"""kernel_stats.csv of tools/lab/general2_prof.py -> achieved GB/s per general2 kernel at (S, B, D) = (94, 30, 200).
Two byte counts per kernel: ALGORITHMIC (every operand once — what an ideal kernel would move to and from HBM) and the
bytes this design really pulls through L2 (one workgroup per (step, dialogue) re-reads its dialogue's S x D operand
matrices: 2 S D floats per workgroup, S B workgroups) — the second is what bounds it (L2-served rate of the chip:
17-18 TB/s, /opt/skills/guides/MI355X_MICROARCH.md)."""
import csv, sys
S, B, D = 94, 30, 200
sbd, bss = 4.0 * S * B * D, 4.0 * B * S * S
alg = {"general2_fwd": 3 * sbd + bss, "general2_bwd_q": 4 * sbd + bss + 4.0 * S * B, "general2_bwd_m": 4 * sbd + bss + 4.0 * S * B}
l2 = 2.0 * S * D * 4 * S * B
for r in csv.DictReader(open(sys.argv[1])):
    for k, b in alg.items():
        if k in r["Name"]:
            us = float(r["AverageNs"]) / 1e3
            print("%-16s calls %4s  avg %7.2f us | algorithmic %.2f MB -> %.0f GB/s (%.1f %% of the 8 TB/s HBM peak) | L2 -> CU "
                  "re-reads by design %.0f MB -> %.1f TB/s (L2-served ceiling 17-18 TB/s)"
                  % (k, r["Calls"], us, b / 1e6, b / us / 1e3, b / us / 1e3 / 80, l2 / 1e6, l2 / us / 1e6))
