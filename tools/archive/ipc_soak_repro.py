#!/usr/bin/env python3
"""Repeats ONE configuration of tools/ipc_soak.py until it fails, every rank reporting its own error and state (diagnosis of an intermittent
wait that gave up around a resumed capacity miss).  usage: ipc_soak_repro.py [world=3] [N=13312] [T=400] [repeats=12] [eager=8]"""
import os, sys, tempfile, traceback
import numpy as np
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))


def rank_main(rank, world, port, n, T, outdir):
    import torch, torch.distributed as dist
    import cases as cs
    from composablestatespacemodels_amd.sharded import DistComm, GpuShard, ShardedFilter
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)

    class GlooPeerComm(DistComm):
        peer = True
    torch.cuda.set_device(0)
    model = cs.c2_model(); t, y, has = cs.poisson_counts(T, missing=0.1)
    shard = GpuShard(model, n, rank, world, cs.SEED, 0); shard.set_option(2, 1)
    f = ShardedFilter([shard], GlooPeerComm())
    cut = T // 3
    try:
        f.ll_filter(t[:cut], y[:cut], has[:cut])
        ll, ess = f.ll_filter_more(t[cut:], y[cut:], has[cut:])
        print(f"rank {rank}: ok ll {ll!r} resumes {f.last_resumes} rows {shard.peer_rows()}", flush=True)
    except Exception as e:
        import ctypes as C
        buf = np.zeros(2 * world * 96 + 257, dtype=np.uint32)
        shard.lib.cssm_pf_shard_peer_debug(shard._h, buf.ctypes.data_as(C.POINTER(C.c_uint32)), buf.size)
        lines = [f"rank {rank}: FAILED {str(e)[:140]} | seq {buf[-1]}"]
        for p in range(2):
            for r in range(world):
                w = buf[(p * world + r) * 96:(p * world + r + 1) * 96]
                ll = sorted(set(int(x) for x in w[33:80:2]))
                lines.append(f"   rank {rank} window {p} from rank {r}: header flag {w[0]} rows {w[16]} need {w[17]} extra {w[18]} LL seqs {ll}")
        tk = buf[2 * world * 96:2 * world * 96 + 256]
        lines.append(f"   rank {rank} tickets A {tk[:world].tolist()} pre_flag {tk[64]} need {tk[128:128 + world].tolist()} tickets B {tk[192:192 + world].tolist()}")
        print("\n".join(lines), flush=True)
        os._exit(3)
    shard.close(); dist.barrier(); dist.destroy_process_group()


if __name__ == "__main__":
    import torch.multiprocessing as mp
    world = int(sys.argv[1]) if len(sys.argv) > 1 else 3
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 13312
    T = int(sys.argv[3]) if len(sys.argv) > 3 else 400
    reps = int(sys.argv[4]) if len(sys.argv) > 4 else 12
    os.environ["CSSM_PEER_EAGER_ROWS"] = sys.argv[5] if len(sys.argv) > 5 else "8"
    os.environ.setdefault("CSSM_GRP_MIN_UNITS", "1"); os.environ.setdefault("CSSM_PEER_TIMEOUT_MS", "3000")
    for i in range(reps):
        print(f"--- repeat {i}", flush=True)
        with tempfile.TemporaryDirectory() as d:
            try:
                mp.spawn(rank_main, args=(world, 29800 + (os.getpid() + i) % 90, n, T, d), nprocs=world, join=True)
            except Exception as e:
                print("spawn:", str(e)[:200]); break
