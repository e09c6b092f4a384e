"""Preflight of the multi-GPU path, the first step of tools/first_8gpu_run.sh: nothing here needs torch.

  1. environment: HSA_ENABLE_IPC_MODE_LEGACY (the host driver only supports dmabuf IPC: RCCL's intra-node transport fails with
     `hipIpcGetMemHandle: invalid argument` without =0), visible devices;
  2. librccl: dlopen + every symbol pysparselp_amd/csrc/slp_comm.hip binds (ncclGetUniqueId, ncclCommInitRank, ncclAllReduce,
     ncclReduceScatter, ncclAllGather, ncclCommDestroy, ncclGetErrorString);
  3. a ONE-rank communicator through the library's own entry points (slp_comm_unique_id / slp_comm_init) and an 80 MB all-reduce
     (n = 1e7 doubles: config 4's exchanged vector, ChambollePockPPD.py:206,216) timed with the library's event pairs;
  4. optionally (--ranks N, N <= devices) the same over N self-launched ranks, one per GPU: the first bytes between two GPUs.

Prints ONE JSON object; exit code 0 only if every step that could run passed.
    python tools/rccl_preflight.py [--ranks N] [--doubles 10000000] [--reps 10]"""
import argparse
import ctypes
import json
import os
import subprocess
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

SYMBOLS = ("ncclGetUniqueId", "ncclCommInitRank", "ncclAllReduce", "ncclReduceScatter", "ncclAllGather", "ncclCommDestroy",
           "ncclGetErrorString")


def rank_main(args):
    """One rank: communicator, `reps` all-reduces of `doubles` doubles (sum), checked; rank 0 prints the timing."""
    import numpy as np

    from pysparselp_amd import _lib
    from pysparselp_amd.parallel import init_comm_from_env

    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    lib = _lib.lib(int(os.environ.get("SLP_DEVICE", os.environ.get("LOCAL_RANK", "0"))))
    init_comm_from_env(rank, world)
    want = world * (world + 1) / 2.0
    # through the host entry point first (small), then the device path the solvers use: a Chambolle-Pock step's exchange
    small = np.array([float(rank + 1)])
    _lib.check(lib.slp_comm_allreduce_host(_lib.ptr(small), 1, 0))
    ok = bool(small[0] == want)
    res = np.zeros(3)
    t0 = time.perf_counter()
    _lib.check(lib.slp_comm_bench_allreduce(args.doubles, args.reps, _lib.ptr(res)))
    dt = time.perf_counter() - t0
    ok = ok and bool(res[1] == 0.0 and int(res[2]) == world)
    _lib.check(lib.slp_comm_barrier())
    _lib.check(lib.slp_comm_finalize())
    if rank == 0:
        print(json.dumps({"ranks": world, "doubles": args.doubles, "reps": args.reps, "sums_correct": ok,
                          "ms_per_allreduce_device_events": float(res[0]), "bytes_per_allreduce": 8 * args.doubles,
                          "algbw_gbps": (8 * args.doubles / 1e9) / (float(res[0]) * 1e-3) if res[0] > 0 else None,
                          "busbw_gbps": ((8 * args.doubles / 1e9) / (float(res[0]) * 1e-3) * 2.0 * (world - 1) / world) if res[0] > 0 and world > 1 else None,
                          "seconds_whole_call": dt}), flush=True)
    return 0 if ok else 1


def main():
    p = argparse.ArgumentParser()
    p.add_argument("--ranks", type=int, default=1)
    p.add_argument("--doubles", type=int, default=10_000_000)
    p.add_argument("--reps", type=int, default=10)
    p.add_argument("--as-rank", action="store_true", help=argparse.SUPPRESS)
    args = p.parse_args()
    if args.as_rank:
        raise SystemExit(rank_main(args))
    out = {"what": "rccl_preflight", "HSA_ENABLE_IPC_MODE_LEGACY": os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY"), "ok": True}
    # 2. the library and its symbols -- before anything initialises a GPU in THIS process (it only launches ranks)
    lib = None
    for name in ("librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so"):
        try:
            lib = ctypes.CDLL(name)
            out["librccl"] = name
            break
        except OSError:
            continue
    if lib is None:
        out.update(ok=False, error="librccl.so cannot be loaded")
        print(json.dumps(out))
        raise SystemExit(1)
    missing = [s for s in SYMBOLS if not hasattr(lib, s)]
    out["symbols_missing"] = missing
    if missing:
        out["ok"] = False
    # device count through a child (slp_device_count does not initialise a context, but keep this process clean anyway)
    r = subprocess.run([sys.executable, "-c", "import sys; sys.path.insert(0, %r)\nfrom pysparselp_amd import _lib\nprint(_lib.load().slp_device_count())" % REPO],
                       capture_output=True, text=True)
    out["devices_visible"] = int(r.stdout.strip() or -1) if r.returncode == 0 else -1
    if out["devices_visible"] < 1:
        out.update(ok=False, error="no HIP device visible")
        print(json.dumps(out))
        raise SystemExit(1)
    # 3. / 4. ranks as child processes (never an exec from a process that touched the GPU)
    import secrets
    import socket

    for world in sorted({1, min(args.ranks, out["devices_visible"])}):
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            port = s.getsockname()[1]
        token = secrets.token_hex(8)
        procs = []
        for rk in range(world):
            env = dict(os.environ, RANK=str(rk), LOCAL_RANK=str(rk), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                       SLP_JOB_TOKEN=token, SLP_FORCE_DISTRIBUTED="1")
            procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__), "--as-rank", "--doubles", str(args.doubles), "--reps",
                                           str(args.reps)], env=env, stdout=subprocess.PIPE if rk == 0 else subprocess.DEVNULL, text=True))
        try:
            line, _ = procs[0].communicate(timeout=600)
            codes = [q.wait(timeout=60) for q in procs]
        except subprocess.TimeoutExpired:
            for q in procs:
                if q.poll() is None:
                    q.kill()
            out[f"allreduce_{world}_ranks"] = {"error": "timeout"}
            out["ok"] = False
            continue
        rec = None
        for ln in (line or "").splitlines():
            if ln.startswith("{"):
                rec = json.loads(ln)
        out[f"allreduce_{world}_ranks"] = rec or {"error": f"exit codes {codes}"}
        if any(codes) or not rec or not rec.get("sums_correct"):
            out["ok"] = False
    print(json.dumps(out))
    raise SystemExit(0 if out["ok"] else 1)


if __name__ == "__main__":
    main()
