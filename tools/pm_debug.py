"""Fused map launch (marius_prepare_maps) under a watchdog: runs the bench batch's two id lists through ONE launch, prints the control block's
give-up records (sort_unique.hip: RsErr) and compares with the separate launches.  MARIUS_PM_NWG=k sets the workgroup count."""
import math
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from marius_amd import hip as H  # noqa: E402


def main():
    import faulthandler
    faulthandler.dump_traceback_later(60, exit=True)
    dev = torch.device("cuda:0")
    B, C, N, num_nodes, R = 50000, 50, 1000, 86054151, 14824
    trials = int(os.environ.get("PM_TRIALS", "3"))
    g = torch.Generator().manual_seed(1)
    CN = C * N
    L = 2 * B + 2 * CN
    nbits, rbits = math.ceil(math.log2(num_nodes)), math.ceil(math.log2(R + 1))
    um, ur, vm, vr = H.UniqueMap(L, dev), H.UniqueMap(B, dev), H.UniqueMap(L, dev), H.UniqueMap(B, dev)
    for trial in range(trials):
        edges = torch.stack([torch.randint(num_nodes, (B,), generator=g), torch.randint(R, (B,), generator=g), torch.randint(num_nodes, (B,), generator=g)], 1).to(dev)
        sneg, dneg = torch.randint(num_nodes, (C, N), generator=g).to(dev), torch.randint(num_nodes, (C, N), generator=g).to(dev)
        ids = torch.empty(L, dtype=torch.int64, device=dev)
        H.check(H.lib().marius_assemble_ids(H.ptr(edges), B, 3, H.ptr(sneg), H.ptr(dneg), CN, H.ptr(ids), H.stream_ptr()), "assemble_ids")
        vm.run(ids, nbits)
        vr.run(edges[:, 1].contiguous(), rbits)
        want_plan, want_rplan = H.segment_plan(vm, L), H.segment_plan(vr, B)
        torch.cuda.synchronize()
        ids_out, rel_out = torch.full((L,), -1, dtype=torch.int64, device=dev), torch.full((B,), -1, dtype=torch.int64, device=dev)
        got_edges = torch.full_like(edges, -1)
        plan = torch.empty(int(H.lib().marius_segment_plan_bytes(L)), dtype=torch.uint8, device=dev)
        rplan = torch.empty(int(H.lib().marius_segment_plan_bytes(B)), dtype=torch.uint8, device=dev)
        jobs = [H.map_job(um, nbits, edges=edges, src_neg=sneg, dst_neg=dneg, ids_out=ids_out, plan=plan, edges_out=got_edges),
                H.map_job(ur, rbits, edges=edges, col=1, ids_out=rel_out, plan=rplan)]
        t0 = time.time()
        print("launching", flush=True)
        assert H.prepare_maps(jobs)
        torch.cuda.synchronize()
        dt = time.time() - t0
        ctl = um.ws[64:256].cpu().view(torch.int32).tolist()
        nerr = ctl[18]
        print("trial %d: %.1f ms  next=%d exited=%d done=%s  nerr=%d" % (trial, dt * 1e3, ctl[0], ctl[1], ctl[2:18], nerr))
        for k in range(min(nerr, 6)):
            w = ctl[19 + 4 * k:23 + 4 * k]
            print("   gave up: where=%d (0 phase wait, 1 sweep look-back, 2 emit look-back) phase=%d tile=%d waited-for=%d seen=0x%x" % (w[0] & 255, w[0] >> 8, w[1], w[2], w[3] & 0xffffffff))
        ok = {"ids": torch.equal(ids_out, ids), "count": int(um.count.item()) == int(vm.count.item()), "uniq": torch.equal(um.uniq[:L], vm.uniq[:L]),
              "inverse": torch.equal(um.inverse[:L], vm.inverse[:L]), "perm": torch.equal(um.perm[:L], vm.perm[:L]),
              "rcount": int(ur.count.item()) == int(vr.count.item()), "rinverse": torch.equal(ur.inverse[:B], vr.inverse[:B]), "rperm": torch.equal(ur.perm[:B], vr.perm[:B])}
        print("   equal to the separate launches:", ok)
        if nerr:
            um.ws[64:256].zero_()
    if os.environ.get("MARIUS_PM_MAXPH"):
        return
    # timing
    st = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    st[0].record()
    for _ in range(50):
        H.prepare_maps(jobs)
    st[1].record()
    torch.cuda.synchronize()
    print("fused: %.1f us per launch" % (st[0].elapsed_time(st[1]) * 20))


if __name__ == "__main__":
    main()
