"""Rank program of tests/test_gpu_dist.py: ShardedSolver with the real HIP engine on every rank.
Launched by torch.distributed.run; the backend is gloo when the ranks have to share one GPU (RCCL
refuses two ranks on one device), nccl when every rank has its own."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    out_path, cfg = sys.argv[1], sys.argv[2]
    rank, world, local = (int(os.environ[k]) for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"))
    ndev = torch.cuda.device_count()
    dev = torch.device("cuda", local % ndev)
    torch.cuda.set_device(dev)
    if ndev >= world:
        dist.init_process_group("nccl", device_id=dev)
    else:
        dist.init_process_group("gloo")
    from cvx_proj_amd import dist as D
    from cvx_proj_amd.dist import ShardedSolver, solve_pairs
    from cvx_proj_amd.synth import config_pair
    if world == 1:
        # one rank on its own device = the nccl backend: do not take the single-process shortcuts, so that RCCL runs every
        # broadcast and all-gather of the multi-rank path once with the real tensors
        D.REHEARSE_ONE_RANK = True
    p = config_pair(cfg, with_image=(rank == 0))
    s = ShardedSolver(p, dev, dist, overlap=True if world == 1 else "auto")
    if rank != 0:
        assert float(s.table.abs().sum()) == 0.0          # the table lives on rank 0 until the broadcast
    H = s.solve().cpu().numpy().copy()
    canvas = s.warp().cpu().numpy().copy()
    H2 = s.solve().cpu().numpy()                          # a second solve does not broadcast again and gives the same grid
    assert np.array_equal(H, H2) and s.overlap
    s_one = ShardedSolver(p, dev, dist, overlap=False)    # one launch + one gather per rank
    assert np.array_equal(s_one.solve().cpu().numpy(), H)
    assert s_one._plan is None                            # a solve-only caller pays for no warp plan (made at the first warp / step)
    s_one = ShardedSolver(p, dev, dist, overlap=False, resident_warp=True)
    assert np.array_equal(s_one.solve().cpu().numpy(), H)
    s_bits = ShardedSolver(p, dev, dist, same_bits=True)  # the whole mesh's summation order on every shard
    H_bits = s_bits.solve().cpu().numpy().copy()
    band = s.warp(gather=False).cpu().numpy()             # the canvas left distributed: this rank's rows only
    lo, hi = s.bands[rank]
    assert np.array_equal(band, canvas[lo:hi])
    assert s._aligned                                     # bands = the canvas rows of the mesh rows this rank solved
    H_step, band_step = s.step()                          # solve -> H gather in flight -> warp of the own band -> wait
    assert np.array_equal(H_step.cpu().numpy(), H) and np.array_equal(band_step.cpu().numpy(), canvas[lo:hi])
    # the resident warp form: one launch per rank whose tail leaves the rank's cells warp ready in its WarpPlan, the warp step
    # is then the gather kernel alone on the rank's band (no set-up launch) - same grid, same bytes
    assert s_one._plan is not None and s_one._cells_ready and s._plan is not None
    H_one, band_one = s_one.step()
    assert s_one._cells_ready
    assert np.array_equal(H_one.cpu().numpy(), H) and np.array_equal(band_one.cpu().numpy(), canvas[lo:hi])
    assert np.array_equal(s_one.warp().cpu().numpy(), canvas) and np.array_equal(s_one.warp(gather=False).cpu().numpy(), canvas[lo:hi])
    assert int(s_one.status.cpu()[0]) == 0
    pairs = [config_pair("C1", with_image=False, seed_offset=k) for k in range(5)]
    grids = solve_pairs(pairs, dev, dist)
    gathered = [None] * world
    dist.all_gather_object(gathered, (H.tobytes(), canvas.tobytes()))
    assert all(g == gathered[0] for g in gathered), "ranks disagree on the gathered grid / canvas"
    if rank == 0:
        np.savez(out_path, H=H, H_same_bits=H_bits, canvas=canvas, parts=np.array(s.parts), bands=np.array(s.bands),
                 status=int(s.status.cpu()[0]), grids=np.stack(grids))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
