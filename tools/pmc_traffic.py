"""Turn two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate passes, MI355X_MICROARCH.md section HBM) of
    python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra-legs --no-profile --no-graph
into profiles/<name>_pmc_traffic.json: HBM-side bytes per launch of the grouped GEMM launches of the round step.
gfx950 correction: FETCH_SIZE tallies 128-B requests at 64 B -> doubled; both counters are in KiB.

    python tools/pmc_traffic.py <fetch_dir> <write_dir> <out.json>
"""
import csv
import glob
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from melissa_amd import build  # noqa: E402  (source hash of the library the passes ran on: bench.py only reports traffic
                                #              from a file whose hash matches the library it is timing)

# launches of the round step by kernel name (the call site is part of it: TAG 1 = conv1, 2 = conv2, 3 = heads)
# (default precision MEL_PREC_F32_AUTO: conv2 and the heads' first layer run the 128 x 128 split-bf16 kernel at this size;
#  the exact-fp32 names stay in the list for builds / workloads that launch those)
LAUNCHES = {"conv1 (lin_l+lin_r, feature tuples)": "gemm_f32_kernel<2, 2, 1, 1, 0>", "conv2 (lin_l+lin_r)": "gemm_planes_kernel<2>",
         "conv2 (lin_l+lin_r), exact fp32": "gemm_f32_persistent_kernel<2, 2, 1, 1, 0, 2>",
         "head0 (Q|V, split-K)": "gemm_split_big_kernel<3>", "head0 (Q|V, split-K), exact fp32": "gemm_f32_ring_kernel<3,",
         "head finish": "head_finish_kernel",
         "conv1 attention": "gat_attend_rows_kernel<8, 0,", "conv2 attention": "gat_attend_rows_kernel<8, 2,",
         "env round": "env_round_kernel", "encoder (feature tuples)": "gemm_f32_kernel<2, 2, 1, 1, 1>",
         "plan lists": "plan_lists_kernel"}


def mean_counter(directory, counter, needle):
    path = glob.glob(directory + "/**/*counter_collection.csv", recursive=True)[0]
    vals = [float(r["Counter_Value"]) for r in csv.DictReader(open(path))
            if r["Counter_Name"] == counter and needle in r["Kernel_Name"]]
    vals = vals[len(vals) // 4:]                 # drop the warm-up launches (env_round: the first call only initialises)
    return sum(vals) / max(len(vals), 1)


def main():
    out = {"command": "rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE (separate passes) -- python3 bench.py --steps 20 --warmup 5 --no-extra-legs "
                      "--no-cpu-baseline --no-profile --no-graph",
           "correction": "hbm_bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024 (gfx950 FETCH_SIZE tallies 128-B requests at 64 B, "
                         "MI355X_MICROARCH.md section HBM)",
           "workload": "L-DGN 50-node, 1024 envs, round loop, fp32 (MEL_PREC_F32_AUTO)", "source_hash": build.source_hash(), "per_launch": {}}
    for name, needle in LAUNCHES.items():
        f, w = mean_counter(sys.argv[1], "FETCH_SIZE", needle), mean_counter(sys.argv[2], "WRITE_SIZE", needle)
        if f == 0 and w == 0:
            continue                                 # this build does not launch that kernel
        out["per_launch"][name] = {"FETCH_SIZE_KB": round(f, 1), "WRITE_SIZE_KB": round(w, 1),
                                   "hbm_bytes_corrected": int((2 * f + w) * 1024)}
    json.dump(out, open(sys.argv[3], "w"), indent=1)
    print(json.dumps(out["per_launch"]))


if __name__ == "__main__":
    main()
