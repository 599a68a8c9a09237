"""kernel_stats.csv of tools/lab/general2_prof.py -> achieved GB/s per general2 kernel at (S, B, D) = (94, 30, 200).
Two byte counts per kernel: ALGORITHMIC (every operand once — what an ideal kernel would move to and from HBM) and the
bytes the dialogue-resident design pulls through L2 (one workgroup per (16 steps, dialogue) stages its dialogue's S x D
memory once; the memory-gradient kernel streams d_att and x of the dialogue once per 16 memory steps)."""
import csv, sys
S, B, D, QB = 94, 30, 200, 16
nblk = (S + QB - 1) // QB
sbd, bss = 4.0 * S * B * D, 4.0 * B * S * S
alg = {"general2_fwd": 3 * sbd + bss, "general2_bwd_q": 4 * sbd + bss + 4.0 * S * B, "general2_bwd_m": 4 * sbd + bss + 4.0 * S * B}
l2 = {"general2_fwd": nblk * B * 4.0 * (S * D + QB * D) + sbd + 2 * bss,
      "general2_bwd_q": nblk * B * 4.0 * (S * D + QB * D) + sbd + 3 * bss,
      "general2_bwd_m": nblk * B * 4.0 * (2 * S * D) + sbd + 2 * bss}
for r in csv.DictReader(open(sys.argv[1])):
    for k, b in alg.items():
        if k in r["Name"]:
            us = float(r["AverageNs"]) / 1e3
            print("%-16s calls %4s  avg %7.2f us | algorithmic %.2f MB -> %.0f GB/s (%.1f %% of the 8 TB/s HBM peak) | through L2 %.1f MB "
                  "-> %.2f TB/s | %d workgroups of 512 threads, one per CU (the S x D image takes half a CU's LDS): "
                  "latency-sized, not bandwidth-sized" % (k, r["Calls"], us, b / 1e6, b / us / 1e3, b / us / 1e3 / 80, l2[k] / 1e6,
                                                          l2[k] / us / 1e6, nblk * B))
