"""Set-up collectives over torch.distributed (any backend, gloo included) for PetscHipInitializeWithCallbacks.

Inside PETSc the same two callbacks are host MPI_Allgather / MPI_Alltoallv; here torch.distributed plays
MPI's role for the handshake of VecScatterCreate_PtoS (vpscat.c:2346-2392) and the MatStash exchange (mpiaij.c:680-737).
No device traffic goes through them.
"""
import numpy as np


def make_callbacks(dist):
    import torch

    world = dist.get_world_size()
    rank = dist.get_rank()

    def allgather(data):
        t = torch.frombuffer(bytearray(data), dtype=torch.uint8) if len(data) else torch.zeros(0, dtype=torch.uint8)
        outs = [torch.empty_like(t) for _ in range(world)]
        if len(data):
            dist.all_gather(outs, t)
        return b"".join(bytes(o.numpy().tobytes()) for o in outs)

    def alltoallv(send, scounts, rcounts):
        # gloo has no all_to_all: gather every rank's (counts, payload) and cut out this rank's share
        lens = [torch.zeros(1, dtype=torch.int64) for _ in range(world)]
        dist.all_gather(lens, torch.tensor([len(send)], dtype=torch.int64))
        mx = max(int(v) for v in lens)
        pad = np.zeros(max(mx, 1), np.int32)
        pad[:len(send)] = send
        bufs = [torch.zeros(max(mx, 1), dtype=torch.int32) for _ in range(world)]
        dist.all_gather(bufs, torch.from_numpy(pad))
        cnts = [torch.zeros(world, dtype=torch.int32) for _ in range(world)]
        dist.all_gather(cnts, torch.from_numpy(np.ascontiguousarray(scounts, np.int32)))
        parts = []
        for p in range(world):
            c = cnts[p].numpy()
            off = int(c[:rank].sum())
            assert int(c[rank]) == int(rcounts[p]), "rank %d sends %d to %d, which expects %d" % (p, c[rank], rank, rcounts[p])
            parts.append(bufs[p].numpy()[off:off + int(c[rank])])
        return np.concatenate(parts) if parts else np.zeros(0, np.int32)

    return allgather, alltoallv
