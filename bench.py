"""Headline benchmark: first-order LP iterations per second on a synthetic random sparse LP.

    python bench.py --gpus N --steps K --warmup W

One "step" = one solver iteration (all SpMV / SpMV^T passes, projections and
multiplier updates of that iteration) on the synthetic random LP of
randomLP.py:29-75, generated on the GPU and resident in HBM before the timed
region.  The default workload is the LP BASELINE.json's metric names: 1e7
variables x 2e7 inequality rows (config 4) at the density that fits one node,
1e-4 (2e10 stored entries; the 1e-3 of config 4 is 2.4 TB of CSR, more than
8 x 288 GB) -- resident on ONE GPU because every rank's row block is generated,
converted and released in row chunks (ChunkedDeviceMatrix: the CSR of the whole
block never exists).  N > 1 row-partitions the SAME problem over the N GPUs
(strong scaling, one RCCL all-reduce of the n partial column sums per SpMV^T):
``python bench.py --gpus N`` starts its own N ranks (``self_launch``: plain child
processes + a TCP id exchange, no torch anywhere); the driver's ``python -m
torch.distributed.run --nproc-per-node N ... bench.py --gpus N`` works too (bench.py
only reads the RANK / WORLD_SIZE / MASTER_* it sets).  ``--config c3`` runs BASELINE config 3 (1e6 x 2e6 at
1e-3, the headline of rounds 1-3); at N = 1 the default run appends it as
``secondary.c3`` (same code path, a few seconds).

Rank 0 prints ONE JSON line (contract in the task statement) carrying

``roofline``      the dominant kernel (the SpMV ``y = A x`` of the timed step): bytes that kernel HAS TO MOVE per launch
                  (the matrix copy it streams + the vector it reads + the vector it writes) / its HIP-event time, against
                  the 8 TB/s HBM peak -- a fraction <= 1 by construction.  The rate in CSR-equivalent algorithmic bytes
                  (12 B per stored entry + vectors), which exceeds the physical rate when the kernel
                  streams the lossless value-dictionary copy (3-4 B per entry), is reported under its own name
                  (``csr_equivalent``), never as ``achieved``.
``roofline.general_fp64``  the same matrix, same run, with the value dictionary ruled out (``slp_matrix_set_format(a, 1)``):
                  fp64 strip entries (10 B per stored entry) -- the path a matrix with arbitrary coefficients takes --
                  SpMV in both orientations, one Chambolle-Pock and one ADMM (reuse level 4) step rate.
``cpu_baseline``  the oracle (CPU restatement of the reference algorithm, single thread like the reference) on a bounded
                  sample of the same workload: the first rows of the same LP (all n columns, so the gathers hit a vector
                  of the full size), ITERATIONS ONLY (setup excluded), scaled by the row ratio; ``extrapolated: true``;
                  ``matrix_products_per_iteration`` = the products of the reference's form of the iteration that it times
                  (ADMM 10, Chambolle-Pock 2; the device spends ``config.matrix_passes_per_iteration`` = 4 / 2 on the same iterates).
``chambolle_pock`` / ``cpu_baseline.chambolle_pock``  (default line) the Chambolle-Pock rate on the same resident LP and the oracle's on
                  the same sample: the method where both sides do the SAME two products per iteration.
``--eq-frac 0.1``  SURVEY 8(d)'s second workload: the first tenth of the rows are equalities a_i x = a_i x_feasible
                  (randomLP.py:62-68); a chunked block is cut at the boundary, Chambolle-Pock keeps the reference's
                  (c + y_eq a_eq) + y_ineq a_ineq bit for bit (csrc/slp_cp.hip cp_split_setup).
"""
import argparse
import json
import math
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)
# multi-process GPU work on this platform: the host driver only supports dmabuf IPC (RCCL's intra-node transport needs it);
# must be in the environment before the HIP runtime initialises
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

# the HIP library first: bench never needs torch on one GPU
from pysparselp_amd import _lib  # noqa: E402
from pysparselp_amd.device import DeviceMatrix  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8 TB/s spec (6.3 TB/s is what a plain copy achieves)

KERNEL_NAMES = {
    7: "k_tall_spmv<false> (tall cells: 10^4-row blocks x 4096-column strips, running sums and x-tile in LDS, 4-byte "
       "(column, row) items + fp64 values)",
    6: "k_tall_spmv (tall cells: 10^4-row blocks x 4096-column strips, running sums and x-tile in LDS, 5-byte "
       "self-describing value-dictionary items, lossless)",
    5: "k_wstrip_spmv<false> (wide strips: x gathered from L2, fp64 entries)",
    4: "k_wstrip_spmv<true> (wide strips: x gathered from L2, value-dictionary entries)",
    3: "k_qstrip_spmv<1> (LDS-tiled strip-JDS SpMV over the value-dictionary copy: 12-bit value id + 12-bit column per "
       "stored entry, lossless)",
    2: "k_dstrip_spmv<1> (LDS-tiled strip-JDS SpMV over the value-dictionary copy: uint16 value id + uint16 column per "
       "stored entry, lossless)",
    1: "k_strip_spmv<0> (LDS-tiled strip-JDS SpMV, fp64 value + uint16 column per stored entry)",
    0: "k_spmv (row-per-lane-group CSR SpMV)",
}
# rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE summaries of this exact workload (tools/profile_c3.sh, tools/summarize_rocprof.py)
PMC_FILES = {   # newest record first
    3: (("r04_c3_pmc_hbm.json", "r03_c3_pmc_hbm.json", "r02_c3_pmc_hbm.json"), "slp::k_qstrip_spmv<1"),   # kernel-name prefixes: the
    2: (("r04_c3_pmc_hbm.json", "r03_c3_pmc_hbm.json", "r02_c3_pmc_hbm.json"), "slp::k_dstrip_spmv<1"),   # template argument lists
    1: (("r04_c3_pmc_hbm.json", "r03_c3_pmc_hbm.json", "r02_c3_pmc_hbm.json"), "slp::k_strip_spmv<0"),    # grew between rounds
}
# named workloads: (variables, rows, density); c4slice = the 1/8 row slice of a 1e7 x 2e7, density-1e-4 LP -- BASELINE
# config 4's per-rank shape at a density that fits (the 1e-3 of config 4 is 2.4 TB of CSR)
# c4 = the metric's named LP, 1e7 x 2e7 at the density that fits one node (1e-4: 2e10 stored entries): this rank's row block is
# generated, converted and released in row chunks (ChunkedDeviceMatrix), so it is resident on 1, 2, 4 or 8 GPUs alike
# c5 = BASELINE config 5: block-splitting ADMM (ADMMBlocks.py) on a 5e7-variable LP at density 1e-4 -- eight row blocks of 5e5
# rows (2e10 stored entries, as many as c4), 8 / N blocks per rank; every block is generated, converted and released on its own
# (DeviceBlocksGroup.from_generator), so the whole LP is resident on ONE GPU as 8 x 26 GB of tall cells
CONFIGS = {"c3": (1_000_000, 2_000_000, 1e-3), "c4slice": (10_000_000, 2_500_000, 1e-4), "c4": (10_000_000, 20_000_000, 1e-4),
           "c5": (50_000_000, 4_000_000, 1e-4)}
CONFIG_BLOCKS = {"c5": 8}   # row blocks of the whole LP (admm_blocks)
# (record, launches of the recorded kernel that make ONE product: round 4 ran a launch per row chunk, round 5 runs one per product)
PMC_FILES_BY_SHAPE = {(10_000_000, 2_500_000, 1e-4): {6: ((("r05_tall_slice_pmc_hbm.json", 1), ("r04_tall_slice_pmc_hbm.json", 1), ("r03_tall_slice_pmc_hbm.json", 1)),
                                                          "slp::k_tall_spmv")},
                      (10_000_000, 20_000_000, 1e-4): {6: ((("r06_c4_pmc_hbm.json", 1), ("r05_c4_pmc_hbm.json", 1), ("r04_c4_pmc_hbm.json", 8)), "slp::k_tall_spmv")},
                      (50_000_000, 4_000_000, 1e-4): {6: ((("r05_c5_pmc_hbm.json", 1), ("r04_c5shape_pmc_hbm.json", 1)), "slp::k_tall_spmv")}}
# A chunk's CSR (12 B per entry) + its conversion temporaries (sorted keys 8 B, pass scratch) sit beside the copies already built:
# at 1.3e9 entries per chunk (config 4: 16 chunks on one GPU) the set-up peaks at 252 of the 309 GB instead of 280 with 8 chunks.
# Since round 5 the chunking no longer shows in the products: all chunks' tall cells run in ONE launch per product (a chunk brings
# its share of a multiple of the CU count of row blocks, slp_matrix_chunked_expect), measured 9.52-9.54 it/s with 16 chunks
# against 9.57 with 8 on one box (profiles/r05_c4_chunks_16_vs_8.txt).
CHUNK_ENTRIES = 1.3e9
UNCHUNKED_ENTRIES = 2.6e9   # a row block up to this size is an ordinary DeviceMatrix (config 3, the 1/8 slice = one of 8 ranks of config 4)


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--config", default=None, choices=sorted(CONFIGS), help="named workload (sets --n / --m / --density)")
    p.add_argument("--steps", type=int, default=20)
    p.add_argument("--warmup", type=int, default=3)
    # (--vars / --rows: spellings that torch.distributed.run's own parser does not mistake for abbreviations of its options)
    p.add_argument("--n", "--vars", dest="n", type=int, default=None, help="variables (default: config c4)")
    p.add_argument("--m", "--rows", dest="m", type=int, default=None, help="inequality constraints")
    p.add_argument("--density", type=float, default=None)
    p.add_argument("--no-secondary", action="store_true", help="skip the config-3 block the default N = 1 run appends")
    p.add_argument("--seed", type=int, default=0)
    p.add_argument("--method", default=None, choices=["admm", "chambolle_pock_ppd", "admm_blocks"],
                   help="default: admm (admm_blocks for --config c5)")
    p.add_argument("--eq-frac", type=float, default=0.0,
                   help="fraction of the constraint rows turned into equalities a_i x = a_i x_feasible (randomLP.py:62-68); "
                        "the default all-inequality LP is the primary workload")
    p.add_argument("--blocks-per-rank", type=int, default=0,
                   help="admm_blocks: row blocks on every rank (DeviceBlocksGroup; default 1, for --config c5: 8 / gpus)")
    p.add_argument("--chunks-per-block", type=int, default=1, help="admm_blocks with block groups: row chunks every block is converted in")
    p.add_argument("--jacobi", action="store_true", help="admm_blocks: Jacobi-preconditioned conjugate gradients (slp_blocks_set_precond)")
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--no-general", action="store_true", help="skip the general (fp64 strip entries) block")
    p.add_argument("--keep-csr", dest="release_csr", action="store_false",
                   help="keep both CSR orientations resident during the timed region (default: released once the strip copies exist)")
    p.add_argument("--cpu-sample-rows", type=int, default=0, help="rows of the CPU sample (default m / 10)")
    p.add_argument("--chunks", type=int, default=0,
                   help="row chunks this rank's block is generated / converted / released in (ChunkedDeviceMatrix); default: as many "
                        "as keep a chunk below 2.6e9 stored entries (1 = an ordinary DeviceMatrix with its CSR)")
    p.add_argument("--format", type=int, default=0, choices=[0, 1, 2],
                   help="slp_matrix_set_format policy of the timed run: 0 best available, 1 no value dictionary, 2 CSR kernels")
    args = p.parse_args()
    args.default_workload = args.config is None and args.n is None and args.m is None and args.density is None
    if args.method is None:
        args.method = "admm_blocks" if args.config == "c5" else "admm"
    args.blocks_total = CONFIG_BLOCKS.get(args.config, 0) if args.method == "admm_blocks" else 0
    if args.blocks_total and not args.blocks_per_rank:
        if args.blocks_total % args.gpus:
            raise SystemExit(f"--config {args.config}: {args.blocks_total} row blocks do not divide over {args.gpus} ranks")
        args.blocks_per_rank = args.blocks_total // args.gpus
    args.blocks_per_rank = max(1, args.blocks_per_rank)
    args.block_group = args.method == "admm_blocks" and (args.blocks_total > 0 or args.blocks_per_rank > 1)
    if args.config or args.default_workload:
        args.n, args.m, args.density = CONFIGS[args.config or "c4"]
    else:  # free shape: what is not given comes from config 3
        d = CONFIGS["c3"]
        args.n, args.m, args.density = args.n or d[0], args.m or d[1], args.density or d[2]
    return args


def device_memory_in_use(lib):
    free, total = np.zeros(1, dtype=np.int64), np.zeros(1, dtype=np.int64)
    _lib.check(lib.slp_device_memory(_lib.ptr(free), _lib.ptr(total)))
    return float(total[0] - free[0]) / 1e9


def spmv_bytes(nnz, rows, cols):
    """algorithmic bytes of y = A x (fp64 values, int32 columns, int64 row pointer)."""
    return 12 * nnz + 8 * (rows + 1) + 8 * cols + 8 * rows


def pmc_traffic(kernel_id, shape):
    """(HBM bytes per PRODUCT from the committed rocprofv3 PMC summary of this workload, its path) or (None, None)."""
    table = PMC_FILES if shape == (1_000_000, 2_000_000, 1e-3) else PMC_FILES_BY_SHAPE.get(shape, {})
    if kernel_id not in table:
        return None, None
    names, kern = table[kernel_id]
    for name in names:
        name, launches = name if isinstance(name, tuple) else (name, 1)
        path = os.path.join(REPO, "profiles", name)
        if os.path.exists(path):
            for full, k in json.load(open(path))["kernels"].items():
                if full.startswith(kern):
                    return k["hbm_bytes_per_launch_corrected"] * launches, "profiles/" + name
    return None, None


def spmv_block(lib, a, transposed, shape, reps=5):
    """Roofline figures of one SpMV orientation of the resident matrix, timed with HIP events on the library's stream.
    Figures are per PRODUCT (y = A x once); a chunked matrix may take several launches for one (``launches_per_product``)."""
    rows, cols = (a.shape[1], a.shape[0]) if transposed else a.shape
    # Average over >= 0.3 s of back-to-back products behind the library's own ~0.1 s warm-up (slp_matrix_bench_spmv): the duration a
    # product has inside the iterations.  Five products right after an idle spell (the host fills the vector) read 5-15 % long on a 3 ms
    # product -- the chip is still coming up to the clocks of a loaded one (config 5's shape: 3.28 ms against 2.80 over 100 products).
    est = a.bench_spmv(transposed, reps=3)
    reps = int(min(400, max(reps, math.ceil(300.0 / max(est, 1e-3)))))
    ms = a.bench_spmv(transposed, reps=reps)
    which = int(lib.slp_matrix_spmv_kernel(a._h, int(transposed)))
    copy_bytes = int(lib.slp_matrix_format_bytes(a._h, int(transposed)))
    moved = copy_bytes + 8 * cols + 8 * rows  # the matrix copy streamed once + x read once + y written once
    alg = spmv_bytes(a.nnz, rows, cols)
    traffic, src = pmc_traffic(which, shape) if not transposed else (None, None)
    launches = max(1, int(lib.slp_matrix_product_launches(a._h, int(transposed))))
    out = {
        "kernel": KERNEL_NAMES.get(which, "?"),
        "achieved": moved / (ms * 1e-3) / 1e9,
        "frac": moved / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
        "ms_per_product": ms,
        "products_timed": reps,
        "launches_per_product": launches,
        "ms_per_launch": ms / launches,
        "bytes_per_product": moved,
        "matrix_copy_bytes_per_product": copy_bytes,
        "csr_equivalent": {"algorithmic_bytes_per_product": alg, "gbps": alg / (ms * 1e-3) / 1e9,
                           "x_hbm_peak": alg / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS},
    }
    if not transposed:
        out["traffic"] = traffic
        out["traffic_source"] = src
    return out, which


def cpu_baseline(args, method, also_cp=False):
    """Oracle (port of the reference algorithm, 1 thread) on a bounded sample: the first m/10 rows of the same LP with all n
    columns (>= 1/10 of the stored entries; per-entry cost on a CPU depends on the size of the gathered vector, which is the
    full one here).  Only iterations are timed (timestamps taken by the oracle's per-iteration hook); the rate is scaled by
    the row ratio -- an extrapolation, validated at full size for both methods by tools/cpu_full_c3.py
    (profiles/r03_cpu_full_c3.json)."""
    from oracle import oracle

    # ~ 2e8 stored entries: 10-30 s of one CPU core for the timed iterations
    rows = args.cpu_sample_rows or max(1, min(args.m // 10, int(2.0e8 / max(args.n * args.density, 1.0))))
    a = DeviceMatrix.random(rows, args.n, args.density, args.seed)
    xf, c, lb, ub, b = a.random_lp_vectors(args.density, args.seed)
    s = a.download()
    a.close()
    def timed(which):
        """(iterations / s on the sample, set-up seconds, iterations run, stamps) of one oracle solver on the resident sample."""
        stamps = []

        def hook(i, *_):
            stamps.append(time.perf_counter())

        t_start = time.perf_counter()
        if which == "chambolle_pock_ppd":
            iters = 8
            oracle.chambolle_pock_ppd(c, None, None, s, None, b, lb, ub, nb_max_iter=iters, nb_iter_plot=10 ** 9, iterate_hook=hook)
        else:
            iters = 5
            oracle.lp_admm_cg(c, None, None, s, None, b, lb, ub, nb_iter=iters - 1, nb_iter_plot=10 ** 9, iterate_hook=hook)
        # stamp k is taken at the same point of iteration k: differences are whole iterations, setup is before stamp 0
        per_iter = (stamps[-1] - stamps[1]) / (len(stamps) - 2)
        return 1.0 / per_iter, stamps[0] - t_start - per_iter, iters, len(stamps)

    # matrix products per iteration as each side forms them: the oracle runs the reference's code shape -- ADMM with the
    # conjugate-gradient x-step (ADMM.py:148,182-201,262 + conjugateGradientLinearSolver.py:36-46) = A^T lambda, four M v =
    # A^T (A v) and A x: 10; Chambolle-Pock (ChambollePockPPD.py:206/216, 235/240) = A^T y and A z: 2 -- while the device
    # rearranges the ADMM iteration into 4 passes over the matrix (reuse level 4, same iterates to rounding) and runs
    # Chambolle-Pock as the same 2
    products_cpu = {"admm": 10, "chambolle_pock_ppd": 2}
    its, setup_s, iters, nstamps = timed(method)
    out = {
        "value": its * rows / args.m,
        "unit": "it/s",
        "cores": 1,
        "kind": "port",
        "extrapolated": True,
        "sample": f"first {rows} of the {args.m} rows of the same LP (all {args.n} columns, density {args.density}: {s.nnz} stored "
                  f"entries), iterations {1}..{nstamps - 1} of {iters} timed without setup: {its:.4f} it/s measured; value = "
                  f"that rate x {rows}/{args.m} (cost per iteration is linear in the rows)",
        "measured_it_per_s_on_sample": its,
        "sample_setup_seconds": setup_s,
        "host_cores_present": os.cpu_count(),
        "matrix_products_per_iteration": products_cpu[method],
        "note_on_product_counts": "the CPU figure times the reference's form of the iteration (matrix_products_per_iteration); the GPU line's "
                                  "config.matrix_passes_per_iteration is what the device's rearranged iteration spends (ADMM: 4 passes for "
                                  "the reference's 10 products; Chambolle-Pock: 2 and 2)",
    }
    if method == "admm" and also_cp:
        # the method where both sides do the SAME two products per iteration, on the sample that is resident anyway
        its_cp, setup_cp, iters_cp, n_cp = timed("chambolle_pock_ppd")
        out["chambolle_pock"] = {"value": its_cp * rows / args.m, "unit": "it/s", "cores": 1, "extrapolated": True,
                                 "measured_it_per_s_on_sample": its_cp, "sample_setup_seconds": setup_cp,
                                 "iterations_timed": n_cp - 2, "matrix_products_per_iteration": products_cpu["chambolle_pock_ppd"]}
    if (args.n, args.m, args.density) == CONFIGS["c4"]:
        # is the row-sample extrapolation valid at n = 1e7?  tools/cpu_sample_scaling.py: the same measurement on 1 %, 2 %, 5 % of the rows
        try:
            rec = json.load(open(os.path.join(REPO, "profiles", "r05_cpu_sample_scaling_c4.json")))
            key = "admm" if method == "admm" else "chambolle_pock_ppd"
            if "chambolle_pock" in out:
                out["chambolle_pock"]["sample_validation"] = {
                    "source": "profiles/r05_cpu_sample_scaling_c4.json",
                    "extrapolated_full_size_it_per_s": {str(r["fraction"]): r["chambolle_pock_ppd"]["extrapolated_full_size_it_per_s"] for r in rec["samples"]},
                    "spread": rec["chambolle_pock_ppd_extrapolations_spread"]}
            out["sample_validation"] = {
                "source": "profiles/r05_cpu_sample_scaling_c4.json",
                "what": "the same oracle timing on 1 % / 2 % / 5 % of the rows at n = 1e7 (1 thread): full-size rates extrapolated from each",
                "extrapolated_full_size_it_per_s": {str(r["fraction"]): r[key]["extrapolated_full_size_it_per_s"] for r in rec["samples"]},
                "spread": rec[key + "_extrapolations_spread"]}
        except Exception:  # optional evidence, never a reason to lose the bench line
            pass
    if (args.n, args.m, args.density) == CONFIGS["c3"]:
        for name in ("r03_cpu_full_c3.json", "r02_cpu_full_c3.json"):  # (r02: Chambolle-Pock only)
            full = os.path.join(REPO, "profiles", name)
            try:
                rec = json.load(open(full)).get(method)
            except Exception:  # the validation record is optional evidence, never a reason to lose the bench line
                rec = None
            if rec:
                out["full_size_validation"] = {"source": "profiles/" + name, **rec}
                # the rate MEASURED at full size (one thread, this workload) is the baseline; the sample's extrapolation
                # (which flatters the CPU: its gathers stay warmer) is kept beside it
                full_rate = rec.get("full_size_it_per_s")
                if full_rate:
                    out["value_extrapolated_from_sample"] = out["value"]
                    out["value"] = float(full_rate)
                    out["extrapolated"] = False
                    out["sample"] = (f"FULL size, one thread, measured by tools/cpu_full_c3.py (profiles/{name}); this run's sample: "
                                     + out["sample"])
                break
    return out


def mem_total_bytes(lib):
    free, total = np.zeros(1, dtype=np.int64), np.zeros(1, dtype=np.int64)
    _lib.check(lib.slp_device_memory(_lib.ptr(free), _lib.ptr(total)))
    return float(total[0])


def exchange_block(args, world, distributed, coll, exch, exch_max, ms_step):
    """How much of a step was exchange.  ``exch`` = slp_comm_timing_read of rank 0, ``exch_max`` the
    max over the ranks of (ms in collectives, slowest collective, ms on the second stream, -ms in collectives)."""
    n_rec = int(exch[0])
    ms_it = float(exch_max[0]) / args.steps
    side_it = float(exch_max[2]) / args.steps
    out = {
        "transport": os.environ.get("SLP_COMM_TRANSPORT", "rccl") if distributed else None,
        "collectives_per_iteration": (coll / args.steps) if distributed else 0,
        "collectives_timed": n_rec,
        "ms_per_iteration": ms_it,                                   # max over the ranks
        "ms_per_iteration_min_over_ranks": -float(exch_max[3]) / args.steps,
        "ms_overlapped_per_iteration": side_it,                      # issued on the second stream beside the next block's projection
        "slowest_collective_ms": float(exch_max[1]),
        "bytes_per_collective": (float(exch[2]) / n_rec) if n_rec else 0.0,
        "algbw_gbps": (float(exch[2]) / 1e9) / (float(exch[1]) * 1e-3) if exch[1] > 0 else None,   # rank 0: payload / time
        "busbw_gbps_allreduce": ((float(exch[2]) / 1e9) / (float(exch[1]) * 1e-3) * 2.0 * (world - 1) / world) if exch[1] > 0 and world > 1 else None,
        "collectives_not_timed": int(exch[5]),
        # serial collectives sit on the compute stream: the step minus them is what the kernels took (the overlapped ones do not
        # extend the step unless they outlast the projection beside them)
        "compute_ms_per_step": ms_step - (ms_it - side_it),
        "fraction_of_step": (ms_it - side_it) / ms_step if ms_step > 0 else None,
    }
    return out


def cpu_baseline_blocks(args, rows_per_block, cg_steps_full):
    """The oracle's matrix-free form of the block iteration (oracle.lp_admm_blocks_cg: ADMMBlocks.py:264-307 with the per-block
    KKT solve of :268-284 done by conjugate gradients -- the reference's sparse LU of a block with 5e5 rows x 5e7 columns cannot
    run) on ONE block cut down to a sample of its rows, all n columns, 1 thread; iterations only.  Scaled to the whole LP by the
    rows (cost per product is linear in them at fixed n) and by the products per block update measured on the GPU at full size
    (the sample's short block is better conditioned and needs fewer CG steps)."""
    from oracle import oracle

    rows = args.cpu_sample_rows or max(1, min(rows_per_block, int(2.5e7 / max(args.n * args.density, 1.0))))
    a = DeviceMatrix.random(rows, args.n, args.density, args.seed)
    xf, c, lb, ub, b = a.random_lp_vectors(args.density, args.seed)
    s = a.download()
    a.close()
    stamps, steps_at = [time.perf_counter()], [0]

    def hook(i, steps_so_far):
        stamps.append(time.perf_counter())
        steps_at.append(steps_so_far)

    hook.wants_steps = True
    iters = 3
    t_start = time.perf_counter()
    oracle.lp_admm_blocks_cg(c, [(oracle.as_csr(s), None, b)], lb, ub, nb_iter=iters, iterate_hook=hook)
    per_iter = (stamps[-1] - stamps[1]) / (len(stamps) - 2)       # iterations 1 .. (warm-started projections, like the timed GPU steps)
    # conjugate-gradient steps of the TIMED iterations only (iteration 0 starts cold and needs the most)
    passes_sample = 3 + 2 * (steps_at[-1] - steps_at[1]) / (len(steps_at) - 2)
    passes_full = 3 + 2 * cg_steps_full
    blocks_total = args.m // rows_per_block
    scale = (rows_per_block / rows) * (passes_full / passes_sample) * blocks_total
    return {
        "value": 1.0 / (per_iter * scale),
        "unit": "it/s",
        "cores": 1,
        "kind": "port",
        "extrapolated": True,
        "form": "matrix-free (conjugate-gradient) per-block projection -- oracle.lp_admm_blocks_cg; the reference's per-block sparse LU "
                "(ADMMBlocks.py:178-243) cannot factorise a 5e5 x 5e7 block",
        "sample": f"ONE block cut to its first {rows} of {rows_per_block} rows (all {args.n} columns, density {args.density}: {s.nnz} stored "
                  f"entries), iterations 1..{iters - 1} of {iters} timed: {per_iter:.3f} s per block update at {passes_sample:.1f} products; "
                  f"value = 1 / (that x {rows_per_block}/{rows} rows x {passes_full:.1f}/{passes_sample:.1f} products per block update "
                  f"(the GPU's count at full size) x {blocks_total} blocks)",
        "seconds_per_block_update_on_sample": per_iter,
        "products_per_block_update_on_sample": passes_sample,
        "products_per_block_update_full_size": passes_full,
        "sample_seconds_total": time.perf_counter() - t_start,
        "host_cores_present": os.cpu_count(),
    }


def timed_steps(lib, solver, warmup, steps):
    solver.iterate(warmup)
    _lib.check(lib.slp_comm_barrier())
    t0 = time.perf_counter()
    solver.iterate(steps)
    _lib.check(lib.slp_comm_barrier())
    return time.perf_counter() - t0


def general_block(lib, args, a, b, c, lb, ub, shape):
    """The same resident matrix with the value dictionary ruled out: fp64 strip entries, the path of a matrix with
    arbitrary coefficients.  Runs AFTER the timed region of the headline number (it rebuilds the strip copies, and the
    ADMM setup without a dictionary row-normalises the matrix in place)."""
    from pysparselp_amd.scale import make_solver

    t0 = time.perf_counter()
    _lib.check(lib.slp_matrix_set_format(a._h, 1))
    ax, which = spmv_block(lib, a, False, shape)
    aty, _ = spmv_block(lib, a, True, shape)
    t_build = time.perf_counter() - t0
    out = {"what": "same matrix, value dictionary ruled out (slp_matrix_set_format(a, 1)): fp64 value per stored entry "
                   "(strips: + uint16 column; tall cells: + 4-byte (column, row) item); the path a matrix with arbitrary "
                   "coefficients takes",
           "spmv": ax, "spmv_transposed": aty, "format_build_and_spmv_seconds": t_build}
    if which not in (1, 7):
        out["note"] = "the matrix does not qualify for an LDS-tiled fp64 format at this size"
    steps = max(5, min(args.steps, 20))
    cp = make_solver("chambolle_pock_ppd", a, b, c, lb, ub)
    dt = timed_steps(lib, cp, 2, steps)
    out["chambolle_pock_it_per_s"] = steps / dt
    out["chambolle_pock_ms_per_step"] = 1e3 * dt / steps
    cp.close()
    admm = make_solver("admm", a, b, c, lb, ub)  # no dictionary: row scaling in place (the matrix is not reused after this)
    dt = timed_steps(lib, admm, 3, steps)
    out["admm_it_per_s"] = steps / dt
    out["admm_ms_per_step"] = 1e3 * dt / steps
    out["admm"] = admm.describe()
    admm.close()
    return out


def self_launch(args):
    """``python bench.py --gpus N`` with N > 1 and no launcher's environment: this process becomes the launcher.  It starts N
    fresh copies of this script -- one rank per GPU, RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT / SLP_JOB_TOKEN
    set, the id exchange on MASTER_PORT + 1 (pysparselp_amd/parallel.py: plain TCP, no torch) -- forwards rank 0's stdout (the
    ONE JSON line) and exits non-zero if any rank does.  Nothing here has touched HIP: the ranks are children of a process
    that never initialised a GPU, and nothing is ever re-executed.  ``python -m torch.distributed.run ... bench.py --gpus N``
    (the driver's line) keeps working: it sets WORLD_SIZE and this function is not reached."""
    import secrets
    import socket
    import subprocess

    def port_block_free(base):   # MASTER_PORT (a launcher's store would sit there), + 1 .. + 8 the id exchange, + 9 the host transport
        socks = []
        try:
            for p in range(base, base + 10):
                s = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
                socks.append(s)
                s.bind(("127.0.0.1", p))
            return True
        except OSError:
            return False
        finally:
            for s in socks:
                s.close()

    base = int(os.environ["MASTER_PORT"]) if "MASTER_PORT" in os.environ else 0
    if not base:
        for _ in range(64):
            base = 20000 + secrets.randbelow(30000)
            if port_block_free(base):
                break
        else:
            raise SystemExit("bench.py: no free block of 10 ports on 127.0.0.1 for the ranks' rendezvous")
    token = os.environ.get("SLP_JOB_TOKEN") or secrets.token_hex(8)
    import signal

    signal.signal(signal.SIGTERM, lambda *_: sys.exit(143))   # (a `timeout` around the launcher ends the ranks too)
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), LOCAL_WORLD_SIZE=str(args.gpus),
                   MASTER_ADDR=os.environ.get("MASTER_ADDR", "127.0.0.1"), MASTER_PORT=str(base), SLP_JOB_TOKEN=token,
                   HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env, cwd=os.getcwd(),
                                      stdout=None if r == 0 else sys.stderr))   # rank 0 prints the line; the others print nothing
    def end_ranks():   # by PID, never by pattern
        for p in procs:
            if p.poll() is None:
                p.terminate()
        for p in procs:
            try:
                p.wait(timeout=20)
            except subprocess.TimeoutExpired:
                p.kill()

    failed = None
    try:
        while failed is None and any(p.poll() is None for p in procs):
            time.sleep(0.2)
            failed = next((p for p in procs if p.poll() not in (None, 0)), None)
    except BaseException:   # the launcher itself is being ended (Ctrl-C, a timeout's SIGTERM as KeyboardInterrupt / SystemExit): no orphans
        end_ranks()
        raise
    if failed is not None:   # a rank died: the others sit in a collective (or its initialisation) for ever -- end them
        time.sleep(2.0)
        end_ranks()
        print(f"bench.py: rank {procs.index(failed)} exited with code {failed.returncode}; job ended", file=sys.stderr, flush=True)
        raise SystemExit(failed.returncode if failed.returncode and failed.returncode > 0 else 1)
    raise SystemExit(0)


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        self_launch(args)   # (never returns)
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("SLP_DEVICE", os.environ.get("LOCAL_RANK", "0")))  # SLP_DEVICE: several ranks on one GPU (tests)
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: start it as `python bench.py --gpus {args.gpus}` (bench.py launches "
                         f"its own ranks) or under a launcher with --nproc-per-node {args.gpus}")
    if args.method == "admm_blocks":
        # short, very wide row blocks: tall row blocks whose strips are shared by several workgroups (partial row sums added
        # in a fixed order; the block ADMM's conjugate gradients have a tolerance bar, slp_tall.hip)
        os.environ.setdefault("SLP_TALL_SPLIT", "-1")
    lib = _lib.lib(local)
    distributed = world > 1 or os.environ.get("SLP_BENCH_FORCE_DIST") == "1"  # the latter: one-GPU test of the N > 1 plumbing
    if distributed:
        from pysparselp_amd.parallel import init_comm_from_env

        init_comm_from_env(rank, world)
    oom = None
    try:
        out = run_workload(lib, args, rank, world, distributed)
    except _lib.SlpError as e:
        # Config 4 / 5 on one GPU peak at ~280 of the device's 309 GB while a chunk is converted.  Should a box offer less, the
        # SAME LP is built from twice as many (half as large) chunks -- results do not depend on the chunking, bit for bit
        # (tests/test_gpu_chunked.py) -- before giving up.
        if world != 1 or "hipMalloc" not in str(e) or args.chunks or args.chunks_per_block > 1:
            raise
        oom = str(e)   # only the message leaves the block: the exception's traceback keeps the failed attempt's frames -- and
        #                through their locals up to ~200 GB of device copies -- alive for as long as `e` is
    if oom is not None:
        import gc

        print(f"bench.py: {oom}; retrying with twice the row chunks", file=sys.stderr, flush=True)
        gc.collect()
        _lib.check(lib.slp_trim())
        if args.block_group:
            args.chunks_per_block = 2
        else:
            args.chunks = 2 * max(1, int(np.ceil(args.m * args.n * args.density / CHUNK_ENTRIES)))
        out = run_workload(lib, args, rank, world, distributed)
        out["config"]["chunks_doubled_after"] = oom
    if rank == 0:
        if world == 1 and args.default_workload and not args.no_secondary and args.method != "admm_blocks":
            # BASELINE config 3 through the same code: the headline workload of rounds 1-3, for continuity.  The headline line
            # does not depend on it: whatever goes wrong here is recorded, the line is printed all the same.
            import copy

            a3 = copy.copy(args)
            a3.n, a3.m, a3.density = CONFIGS["c3"]
            a3.chunks = 0
            a3.no_cpu_baseline = True
            try:
                sec = run_workload(lib, a3, rank, world, distributed)
                out["secondary"] = {"c3": {k: sec[k] for k in ("metric", "value", "unit", "ms_per_step", "config", "roofline",
                                                               "objective_after_run", "setup_seconds", "setup_breakdown", "device_memory")}}
            except Exception as e:  # noqa: BLE001
                out["secondary"] = {"c3": {"error": f"{type(e).__name__}: {e}"}}
        print(json.dumps(out), flush=True)
    if distributed:
        _lib.check(lib.slp_comm_barrier())
        _lib.check(lib.slp_comm_finalize())


def run_workload(lib, args, rank, world, distributed):
    """Generates this rank's row block of the LP ``args`` names, sets the solver up, times ``args.steps`` iterations between two
    barriers; rank 0 returns the JSON line's dict (the other ranks ``None``).  Everything is released before returning."""
    # ---- the workload: this rank's row block, generated in HBM
    from pysparselp_amd.parallel import row_block

    shape = (args.n, args.m, args.density)
    r0, rows = row_block(args.m, world, rank)
    entries = rows * args.n * args.density
    chunks = args.chunks or (1 if entries <= UNCHUNKED_ENTRIES else int(np.ceil(entries / CHUNK_ENTRIES)))
    alloc = np.zeros(5)
    _lib.check(lib.slp_alloc_stats(None, 1))
    t_gen = time.perf_counter()
    from pysparselp_amd.problems import random_lp_on_device

    from pysparselp_amd.scale import DeviceBlocksGroup, make_solver

    if args.block_group:
        # block-splitting ADMM over several row blocks of this rank: every block generated from its own row range, converted,
        # its CSR released before the next one is generated -- no resident matrix to cut the blocks out of
        chunks = args.chunks_per_block
        cuts = [rows * g // args.blocks_per_rank for g in range(args.blocks_per_rank + 1)]
        solver, xf, c, lb, ub, b = DeviceBlocksGroup.from_generator(args.n, args.m, args.density, args.seed, cuts, row_offset=r0,
                                                                    chunks_per_block=chunks)
        a, owns_a = solver._mats[0], False   # the products are timed on the first block
        nnz_local = solver.nnz
        t_generate = solver.seconds_generating
    else:
        m_eq_local = 0
        if args.eq_frac > 0:  # the first eq_frac * m GLOBAL rows are equalities b_eq = A_e x_feasible (randomLP.py:62-68); this rank
            #                   holds its share; a chunked block is cut at the boundary between the two kinds of rows
            m_eq_global = int(round(args.eq_frac * args.m)) & ~1
            m_eq_local = max(0, min(rows, m_eq_global - r0))
        a, xf, c, lb, ub, b = random_lp_on_device(args.n, args.m, args.density, seed=args.seed, row_offset=r0, rows=rows, chunks=chunks,
                                                  m_eq=m_eq_local)
        owns_a = True
        nnz_local = a.nnz
        _lib.check(lib.slp_synchronize())
        t_generate = time.perf_counter() - t_gen  # the synthetic LP itself (randomLP.py's part); the rest of setup_seconds is the solver's
        if args.format:
            _lib.check(lib.slp_matrix_set_format(a._h, args.format))
        solver = make_solver(args.method, a, b, c, lb, ub, m_eq=m_eq_local, jacobi=args.jacobi)
    _lib.check(lib.slp_synchronize())
    t_gen = time.perf_counter() - t_gen
    # Steady state keeps only what the iteration reads: when both orientations run on strip copies, the two CSR copies
    # (48 GB at config 3) are dropped and the cached temporaries of the setup returned to the driver.
    _lib.check(lib.slp_alloc_stats(_lib.ptr(alloc), 0))
    mem = {"in_use_after_setup_gb": device_memory_in_use(lib), "peak_held_by_the_library_gb": alloc[1] / 1e9}
    released = args.release_csr and not args.block_group and chunks == 1 and a.spmv_kernel(False) >= 1 and a.spmv_kernel(True) >= 1
    if released:
        a.release_csr()
    _lib.check(lib.slp_trim())
    mem["in_use_in_timed_region_gb"] = device_memory_in_use(lib)
    mem["csr_released"] = bool(released)

    # ---- timed region: W warm-up steps, then exactly K steps between two barriers
    solver.iterate(args.warmup)
    cg0 = solver.cg_steps() if args.method == "admm_blocks" else 0
    _lib.check(lib.slp_comm_barrier())
    coll0 = int(lib.slp_comm_collectives())
    _lib.check(lib.slp_comm_timing(1))   # HIP event pairs around every collective: recorded now, read after the timed region
    _lib.check(lib.slp_product_timing(1))  # ... and around every product of the solver (the stream it runs on): roofline.timed_region
    t0 = time.perf_counter()
    solver.iterate(args.steps)
    _lib.check(lib.slp_comm_timing(0))   # (the closing barrier's all-reduce is not part of an iteration)
    _lib.check(lib.slp_product_timing(0))
    _lib.check(lib.slp_comm_barrier())
    dt = time.perf_counter() - t0
    prod = np.zeros(3)
    _lib.check(lib.slp_product_timing_read(_lib.ptr(prod)))   # products, sum of their durations (ms), the longest (ms)
    coll = int(lib.slp_comm_collectives()) - coll0 - 1  # the closing barrier is one
    tmax = np.array([dt])
    _lib.check(lib.slp_comm_allreduce_host(_lib.ptr(tmax), 1, 1))
    dt = float(tmax[0])
    exch = np.zeros(6)
    _lib.check(lib.slp_comm_timing_read(_lib.ptr(exch)))
    exch_max = np.array([exch[1], exch[3], exch[4], -exch[1]])   # ms in collectives (max / -min over the ranks), slowest one, overlapped part
    _lib.check(lib.slp_comm_allreduce_host(_lib.ptr(exch_max), 4, 1))

    nnz = np.array([float(nnz_local)])
    _lib.check(lib.slp_comm_allreduce_host(_lib.ptr(nnz), 1, 0))
    nnz_total = int(nnz[0])
    obj = solver.objective()

    out = None
    if rank == 0:
        ms_step = 1e3 * dt / args.steps
        ax, which = spmv_block(lib, a, False, shape)
        aty, _ = spmv_block(lib, a, True, shape)
        passes = solver.matrix_passes_per_iteration()
        if args.method == "admm_blocks":  # per block: 3 products + 2 per conjugate-gradient step (rank 0's count over its blocks)
            passes = 3 * args.blocks_per_rank + 2 * (solver.cg_steps() - cg0) / args.steps
            if args.block_group:   # the products were timed on ONE block: bytes per iteration = passes x that block's bytes
                pass
        # one iteration = `passes` single-vector sweeps, alternating orientations: bytes the iteration has to move
        # (block groups: every block of the rank runs its own `passes`; the products are timed on the first block)
        iter_bytes = passes / 2.0 * (ax["bytes_per_product"] + aty["bytes_per_product"])
        roofline = {
            "bound": "hbm",
            "kernel": ax["kernel"] + " -- rank 0's row block",
            "achieved": ax["achieved"],
            "peak": HBM_PEAK_GBS,
            "unit": "GB/s",
            "frac": ax["frac"],
            "traffic": ax["traffic"],
            "traffic_source": ax["traffic_source"],
            "traffic_measured_in_this_run": False,  # a committed rocprofv3 PMC summary of this command, not a counter of this run
            "definition": "achieved = bytes the kernel has to move per product y = A x (the matrix copy it streams + x read once + y "
                          "written once) / HIP-event time per product on the library's stream; traffic = rocprofv3 PMC "
                          "(FETCH_SIZE x 2 + WRITE_SIZE, separate passes) of the same kernel on the same workload, per product",
            "bytes_per_product": ax["bytes_per_product"],
            "matrix_copy_bytes_per_product": ax["matrix_copy_bytes_per_product"],
            "ms_per_product": ax["ms_per_product"],
            "products_timed": ax["products_timed"],               # back-to-back, behind ~0.1 s of untimed ones (spmv_block)
            # the products of the K timed steps themselves (HIP event pairs around each, both orientations: the solver's own
            # launches on the stream they ran on): their number, average duration and what that is in bytes moved per second
            "timed_region": (None if prod[0] == 0 else {
                "products": int(prod[0]), "products_per_step": prod[0] / args.steps,
                "ms_per_product": prod[1] / prod[0], "longest_ms": prod[2],
                "bytes_per_product": 0.5 * (ax["bytes_per_product"] + aty["bytes_per_product"]),
                "achieved": 0.5 * (ax["bytes_per_product"] + aty["bytes_per_product"]) / (prod[1] / prod[0] * 1e-3) / 1e9,
                "frac": 0.5 * (ax["bytes_per_product"] + aty["bytes_per_product"]) / (prod[1] / prod[0] * 1e-3) / 1e9 / HBM_PEAK_GBS,
                "share_of_step": prod[1] / (1e3 * dt)}),
            "launches_per_product": ax["launches_per_product"],   # a chunked matrix may take one launch per row chunk
            "ms_per_launch": ax["ms_per_launch"],                 # = ms_per_product / launches_per_product
            "csr_equivalent": ax["csr_equivalent"],
            "spmv_transposed": aty,
            "iteration": {"bytes": iter_bytes, "achieved": iter_bytes / (ms_step * 1e-3) / 1e9,
                          "frac_of_all_gpus": iter_bytes / (ms_step * 1e-3) / 1e9 / (HBM_PEAK_GBS * world),
                          "csr_equivalent_gbps": passes * 12 * nnz_total / (ms_step * 1e-3) / 1e9},
        }
        out = {
            "metric": {"admm": "admm", "admm_blocks": "admm_blocks", "chambolle_pock_ppd": "chambolle_pock"}[args.method] + "_iterations_per_sec",
            "value": args.steps / dt,
            "unit": "it/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": ms_step,
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {
                "workload": f"randomLP synthetic: {args.n} vars, {args.m} "
                            + (f"rows (the first {int(round(args.eq_frac * args.m)) & ~1} equalities, the others inequalities)" if args.eq_frac > 0
                               else "inequality rows") + f", density {args.density}, "
                            f"{nnz_total} stored entries, method {args.method} ({solver.describe()}), "
                            f"rows partitioned over {world} GPU(s)"
                            + (f", {args.blocks_per_rank} row block(s) of the block-splitting ADMM per rank, each generated from its own "
                               "row range, converted and its CSR released before the next (no resident matrix)" if args.block_group else "")
                            + (f", every rank's block built in {chunks} row chunks whose CSR never coexists" if chunks > 1 else "")
                            + (f", SLP_TALL_SPLIT={os.environ['SLP_TALL_SPLIT']} (strip ranges of a tall row block shared by several "
                               "workgroups, partial sums added in range order)" if os.environ.get("SLP_TALL_SPLIT") else ""),
                "chunks_per_rank": chunks,
                "n": args.n, "m": args.m, "density": args.density, "seed": args.seed, "nnz": nnz_total, "eq_frac": args.eq_frac,
                "equality_rows": (int(round(args.eq_frac * args.m)) & ~1) if args.eq_frac > 0 else 0,
                "method": args.method, "matrix_passes_per_iteration": passes,
                "collectives_per_iteration": (coll / args.steps) if distributed else 0,
                **({"cg_steps_per_iteration": (solver.cg_steps() - cg0) / args.steps, "jacobi": bool(args.jacobi),
                    "blocks_per_rank": args.blocks_per_rank, "blocks_total": args.blocks_per_rank * world,
                    "cg_steps_per_block_update": (solver.cg_steps() - cg0) / args.steps / args.blocks_per_rank}
                   if args.method == "admm_blocks" else {}),
                "shard_updates": os.environ.get("SLP_SHARD_UPDATES") == "1",
            },
            # what of a step was exchange: HIP event pairs around every collective of the timed region (slp_comm_timing), max over
            # the ranks; under RCCL a pair brackets the collective's kernel, whose duration includes the wait for the slowest peer
            "exchange": exchange_block(args, world, distributed, coll, exch, exch_max, ms_step),
            "roofline": roofline,
            "objective_after_run": obj,
            "setup_seconds": t_gen,
            "setup_breakdown": {"generate_lp_seconds": t_generate, "solver_setup_seconds": t_gen - t_generate,
                                "allocation_seconds": alloc[0], "allocation_seconds_hidden_on_a_helper_thread": alloc[4],
                                "driver_allocation_calls": int(alloc[3]),
                                "peak_device_gb": alloc[1] / 1e9,
                                **({"note": "generate_lp_seconds includes every chunk's conversion into its product copies"}
                                   if chunks > 1 else {})},
            "device_memory": mem,
        }
        solver.close()
        solver = None
        if world == 1 and args.method == "admm" and args.default_workload and not args.no_cpu_baseline:
            # Chambolle-Pock on the same resident LP, beside the ADMM figure: the method where the CPU path and the device do the SAME
            # two products per iteration (cpu_baseline.chambolle_pock is its partner)
            try:
                cp = make_solver("chambolle_pock_ppd", a, b, c, lb, ub, m_eq=m_eq_local)
                dt_cp = timed_steps(lib, cp, 2, args.steps)
                out["chambolle_pock"] = {"value": args.steps / dt_cp, "unit": "it/s", "ms_per_step": 1e3 * dt_cp / args.steps,
                                         "steps": args.steps, "matrix_passes_per_iteration": cp.matrix_passes_per_iteration(),
                                         "objective_after_run": cp.objective()}
                cp.close()
            except Exception as e:  # noqa: BLE001 -- never a reason to lose the headline line
                out["chambolle_pock"] = {"error": f"{type(e).__name__}: {e}"}
        if world == 1 and not args.no_general and args.format == 0 and which >= 2 and args.method != "admm_blocks" and chunks == 1:
            if released:  # the CSR entries are gone: the same rows again from the counter-based generator
                a.close()
                a = DeviceMatrix.random(rows, args.n, args.density, args.seed, r0)
            roofline["general_fp64"] = general_block(lib, args, a, b, c, lb, ub, shape)
        if world == 1 and not args.no_general and args.format == 0 and which == 6 and args.method != "admm_blocks" and chunks > 1:
            # The metric's LP with ARBITRARY coefficients: fp64 entries are 12.4 B per stored entry as tall cells -- both
            # orientations of the whole LP would be 2.4 x the 5.2 B dictionary copies (509 GB) and do not fit one GPU.  What one
            # GPU holds of it is measured instead: the first EIGHTH of the rows (the row block of one of 8 ranks = --config
            # c4slice) with the dictionary ruled out.
            a.close()
            _lib.check(lib.slp_trim())
            part = (rows // 8) & ~1
            a = DeviceMatrix.random(part, args.n, args.density, args.seed, r0)
            share = general_block(lib, args, a, b[:part], c, lb, ub, shape)
            fp64_bytes = share["spmv"]["matrix_copy_bytes_per_product"] + share["spmv_transposed"]["matrix_copy_bytes_per_product"]
            fp64_bytes *= rows / part
            share.update({"resident_on_one_gpu": False, "share_of_rows": part / rows,
                          "fp64_copies_of_the_whole_lp_gb": fp64_bytes / 1e9,
                          "min_gpus": int(np.ceil(fp64_bytes / (0.9 * mem_total_bytes(lib)))),
                          "note": f"measured on the first eighth of the rows ({part} rows: the row block of one of 8 ranks); the whole LP "
                                  "with fp64 entries needs min_gpus GPUs, where a rank's rates are these per-share rates scaled by its "
                                  "share of the rows (plus the exchange)"})
            roofline["general_fp64"] = share
        if world == 1 and not args.no_cpu_baseline:
            a.close()
            a = None
            _lib.check(lib.slp_trim())
            if args.method == "admm_blocks":
                # (the reference's per-block sparse LU of a KKT matrix with 5e5+ unknowns does not finish in bench time even on a
                # sample: the oracle's matrix-free form of the same iteration is timed, oracle.lp_admm_blocks_cg)
                out["cpu_baseline"] = cpu_baseline_blocks(args, rows // args.blocks_per_rank, out["config"]["cg_steps_per_block_update"])
            else:
                out["cpu_baseline"] = cpu_baseline(args, args.method, also_cp="chambolle_pock" in out)
    if solver is not None:
        solver.close()
    if a is not None:
        a.close()
    _lib.check(lib.slp_trim())
    return out if rank == 0 else None


if __name__ == "__main__":
    main()
