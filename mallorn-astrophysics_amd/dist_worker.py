"""One rank of ``dist.extract_multi_gpu``: started by ``torch.distributed.run`` (argv: batch directory, feature-set
mask, backend).  Selects its GPU BEFORE anything else touches the runtime, maps the batch files, runs its shard through
``dist.extract_sharded`` and -- on rank 0 -- writes ``out.npy`` (+ ``status.npy``) into the batch directory."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def main():
    tmp, mask, backend = sys.argv[1], int(sys.argv[2]), sys.argv[3]
    import numpy as np
    import torch
    import torch.distributed as dist

    rank = int(os.environ["RANK"])
    world = int(os.environ["WORLD_SIZE"])
    local = int(os.environ.get("LOCAL_RANK", rank))
    ndev = torch.cuda.device_count()
    if ndev < 1:
        raise SystemExit("lcfe dist worker: no HIP device visible (lcfe has no CPU fallback)")
    if backend == "nccl" and ndev < world:
        raise SystemExit(f"lcfe dist worker: backend nccl needs one GPU per rank ({world} ranks, {ndev} devices)")
    dev = local % ndev
    torch.cuda.set_device(dev)
    if backend == "nccl":
        dist.init_process_group("nccl", device_id=torch.device("cuda", dev))
    else:
        dist.init_process_group(backend)
    from mallorn_astrophysics_amd.dist import extract_sharded

    csr = {k: np.load(os.path.join(tmp, f"{k}.npy"), mmap_mode="r") for k in ("offsets", "t", "flux", "err", "band")}
    zp = os.path.join(tmp, "z.npy")
    z = np.load(zp, mmap_mode="r") if os.path.exists(zp) else None
    res = extract_sharded(mask, csr, z, return_status=True)
    if rank == 0:
        out, status = res
        if status is not None:
            np.save(os.path.join(tmp, "status.npy"), status)
        np.save(os.path.join(tmp, "out.npy"), out)             # written last: its presence means the run is complete
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
