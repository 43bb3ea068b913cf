"""Child of tests/test_gpu_contract.py::test_frame_parallel_hip_engine_two_ranks: one rank of
parallel.FrameParallelClip on the REAL HIP engine. Launched with torch.distributed.run; the
ranks share the box's one GPU, so the process group is gloo (host-staged halo / gather) -- the
engine, the shard arithmetic, the halo import/export and the interior/edge head split are the
product code paths; only the transport differs from RCCL. A fifth argument names the backend: "nccl"
(= RCCL) is what tests/test_gpu_contract.py::test_rccl_world_size_one_runs_the_device_collectives passes, with one
rank (one GPU per rank is RCCL's rule), so that init_process_group("nccl", device_id=...), the device-tensor broadcast of
the weight arena and the device all_gather_into_tensor of the records execute at least once on this pool."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

import numpy as np
import torch
import torch.distributed as dist

from playaid_core_amd import parallel, synth
from playaid_core_amd.engine import Engine


def main():
    n_total, h, w, out_path = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
    backend = sys.argv[5] if len(sys.argv) > 5 else "gloo"
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    device = torch.device("cuda", 0)
    torch.cuda.set_device(device)
    if backend == "nccl":
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group("nccl", device_id=device)   # as bench.py does for N > 1
    else:
        dist.init_process_group(backend)
    rank, world = dist.get_rank(), dist.get_world_size()
    assert parallel._host_staged() == (backend != "nccl")
    sd = synth.make_state_dict(seed=1234) if rank == 0 else None

    def make_engine(weights):
        return Engine(weights, device="cuda:0", max_batch_frames=16, max_clip_frames=max(n_total, 64),
                      max_frame_height=h, max_frame_width=w)

    eng = parallel.broadcast_engine(make_engine, sd, device)  # rank 1 adopts rank 0's weight arena
    lo, hi = parallel.shard_range(n_total, world, rank)
    frames = synth.make_frames_torch(hi - lo, h, w, first_frame=lo, device=device)
    boxes = torch.from_numpy(synth.make_boxes(hi - lo, h, w, first_frame=lo)).to(device)
    runner = parallel.FrameParallelClip(eng, 7, 3, collectives_at_world_one=True)   # (one rank: the record gather runs all the same)
    outs = []
    for pipeline in (False, True, True):
        rec, lp = runner.run(frames, boxes, n_total, gather=True, pipeline=pipeline)
        outs.append((rec.cpu().numpy(), lp.cpu().numpy()))
    if rank == 0:
        np.savez(out_path, rec=outs[0][0], logp=outs[0][1], rec_p=outs[2][0], logp_p=outs[2][1],
                 interior=np.array([parallel.interior_frame_nums(n_total, world, r, 27) for r in range(world)]),
                 backend=np.array(dist.get_backend()), arena_on_device=np.array(not parallel._host_staged()))
    dist.barrier()
    dist.destroy_process_group()
    eng.close()


if __name__ == "__main__":
    main()
