"""kernel_stats.csv of tools/lab/general2_prof.py -> achieved GB/s per general2 kernel at (S, B, D) = (94, 30, 200).
Algorithmic bytes (every operand once): forward reads x, mem [S x B x D] and writes att [S x B x D], alpha [B x S x S];
backward-q reads d_att, x, mem, alpha and writes dx (+ the per-query scalars); backward-m reads d_att, x, alpha, the
scalars and writes dmem."""
import csv, sys
S, B, D = 94, 30, 200
sbd, bss = 4.0 * S * B * D, 4.0 * B * S * S
alg = {"general2_fwd": 3 * sbd + bss, "general2_bwd_q": 4 * sbd + bss + 4.0 * S * B, "general2_bwd_m": 4 * sbd + bss + 4.0 * S * B}
for r in csv.DictReader(open(sys.argv[1])):
    for k, b in alg.items():
        if k in r["Name"]:
            us = float(r["AverageNs"]) / 1e3
            print("%-16s calls %4s  avg %7.2f us  algorithmic %.2f MB  -> %.0f GB/s (%.1f %% of 8 TB/s HBM; operands are "
                  "L2-resident at this size, the kernel is launch-latency-sized)" % (k, r["Calls"], us, b / 1e6, b / us / 1e3, b / us / 1e3 / 80))
