#!/usr/bin/env python3
"""HIP-IPC export probe (round 5): which allocator situations make `storage._share_cuda_()` fail with
`hipIpcGetMemHandle: invalid argument` on this pool?  Each case exports an 80 MB tensor (the size of the parameter arena),
and reports ok / the error (export only: `hipIpcGetMemHandle` is the call that failed in the topology driver; the import side is
exercised by tests/test_topology_gpu.py).

    python tools/dbg/ipc_probe.py [hold_gb]

hold_gb: gigabytes this process keeps allocated meanwhile (the pytest process in front of the topology driver holds some).
Cases:
  fresh        tensor in its own, exactly sized segment (first allocation of the process)
  sub_big      tensor carved out of a cached multi-GB segment (allocate 6 GB, free, allocate 80 MB)
  sub_mid      tensor carved out of a cached 400 MB segment
  reexport     export, child imports and exits, free, re-allocate the SAME block, export again (x5)
  pool         tensor from a private torch.cuda.MemPool(no_split=True): its own hipMalloc whatever the cache holds (x5, after
               the cache has been filled and fragmented by the cases above)
"""
import json
import sys

import torch
import torch.multiprocessing as mp

N = 20 * 1024 * 1024          # floats: 80 MB


def child(t, q):
    t.add_(1.0)
    torch.cuda.synchronize()
    q.put(float(t[0].item()))


def export(t, with_child=False):
    try:
        t.untyped_storage()._share_cuda_()
    except Exception as e:      # noqa: BLE001
        return "EXPORT FAILED: %s" % (str(e).splitlines()[0][:160],)
    if not with_child:
        return "ok (export only)"
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=child, args=(t, q))
    try:
        p.start()
        v = q.get(timeout=120)
        p.join(60)
    except Exception as e:      # noqa: BLE001
        return "CHILD FAILED: %r" % (e,)
    torch.cuda.synchronize()
    return "ok" if (v == 1.0 and float(t[0].item()) == 1.0 and p.exitcode == 0) else "WRONG value %s exit %s" % (v, p.exitcode)


def main():
    hold_gb = float(sys.argv[1]) if len(sys.argv) > 1 else 0.0
    res = {"hold_gb": hold_gb}
    dev = torch.device("cuda:0")
    hold = [torch.empty(int(1024 ** 3), dtype=torch.uint8, device=dev) for _ in range(int(hold_gb))]
    t = torch.zeros(N, device=dev)
    res["fresh"] = export(t)
    big = torch.empty(6 * 1024 ** 3, dtype=torch.uint8, device=dev)
    del big
    t2 = torch.zeros(N, device=dev)
    res["sub_big"] = export(t2)
    res["sub_big_segment_MB"] = max(s["total_size"] for s in torch.cuda.memory_snapshot()) / 2 ** 20
    mid = torch.empty(400 * 1024 ** 2, dtype=torch.uint8, device=dev)
    del mid
    t3 = torch.zeros(N, device=dev)
    res["sub_mid"] = export(t3)
    out = []
    for i in range(5):
        t4 = torch.zeros(N, device=dev)
        ptr = t4.data_ptr()
        out.append((export(t4), hex(ptr)))
        del t4
        torch.cuda.ipc_collect()
    res["reexport"] = out
    pool = torch.cuda.MemPool(no_split=True)
    out = []
    keep = []
    for i in range(5):
        with torch.cuda.use_mem_pool(pool):
            t5 = torch.zeros(N, device=dev)
        seg = [s for s in torch.cuda.memory_snapshot() if s["address"] <= t5.data_ptr() < s["address"] + s["total_size"]]
        out.append((export(t5), seg[0]["total_size"] / 2 ** 20 if seg else None))
        keep.append(t5)
    res["pool"] = out
    del hold
    print("IPC_PROBE " + json.dumps(res), flush=True)


if __name__ == "__main__":
    main()
