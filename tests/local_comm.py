"""Test infrastructure: all R shards of a sharded filter inside ONE process, the "collectives" done by tensor copies -- how the
stage kernels of composablestatespacemodels_amd.sharded are exercised at world > 1 on a single GPU (tests/test_gpu_sharded.py)
and, with the oracle shard, on the CPU (tests/test_sharded_gloo.py).  Not a performance path, not part of the package."""
import numpy as np
import torch


class LocalComm:
    """All R shards live in this process; the "collectives" are tensor copies.  Test vehicle for the
    stage kernels at world > 1 on a single GPU -- not a performance path."""

    def __init__(self, world: int):
        self.world, self.rank = world, 0

    def all_gather(self, outs, ins):
        cat = torch.cat([i.reshape(-1) for i in ins])
        for o in outs:
            o.copy_(cat)

    def all_to_all_counts(self, outs, ins):
        for q, o in enumerate(outs):
            for r, i in enumerate(ins):
                o[r] = i[q]

    def all_to_all_v(self, outs, ins, out_splits, in_splits):
        R = self.world
        in_off = [np.concatenate([[0], np.cumsum(in_splits[r])]) for r in range(R)]
        for q in range(R):
            pos = 0
            for r in range(R):
                n = int(in_splits[r][q])
                assert n == int(out_splits[q][r])
                if n:
                    outs[q][pos:pos + n].copy_(ins[r][int(in_off[r][q]):int(in_off[r][q]) + n])
                pos += n

    def all_to_all_equal(self, outs, ins):
        R = self.world
        seg = ins[0].numel() // R
        for q in range(R):
            for r in range(R):
                outs[q][r * seg:(r + 1) * seg].copy_(ins[r][q * seg:(q + 1) * seg])

    def all_reduce_sum(self, tensors):
        total = tensors[0].clone()
        for x in tensors[1:]:
            total += x.to(total.device)
        for x in tensors:
            x.copy_(total)

    def agree_max(self, values):
        return max(int(v) for v in values)

    def combine_rows(self, arrays):
        out = np.zeros(arrays[0].shape, dtype=np.uint64)
        for a in arrays:
            out |= np.ascontiguousarray(a).view(np.uint64).reshape(out.shape)
        return out.view(np.float64)

    def barrier(self):
        pass


class LocalCommTrimmed(LocalComm):
    """LocalComm whose equal-split all-to-all moves only what the trimmed all-to-all-v of the library would move (whole
    segments between adjacent ranks, ``header_words`` doubles between every other pair and to oneself) and fills the rest
    of every receive segment with NaN: a kernel that read anything else of a non-adjacent segment could not produce the
    oracle's bits.  Test vehicle (tests/test_gpu_sharded.py), world >= 3."""

    def __init__(self, world: int, header_words: int = 12):
        super().__init__(world)
        self.header_words = header_words

    def all_to_all_equal(self, outs, ins):
        R = self.world
        seg = ins[0].numel() // R
        for q in range(R):
            outs[q].fill_(float("nan"))
            for r in range(R):
                n = seg if abs(q - r) == 1 else min(self.header_words, seg)
                outs[q][r * seg:r * seg + n].copy_(ins[r][q * seg:q * seg + n])


class LocalCommPeer(LocalComm):
    """LocalComm on the PEER-WRITTEN exchange: the shards of this process map each other's receive windows by plain pointers
    (cssm_peer_handle.local_ptr) and the ordinary exchanges of a series are k_boundary_pack writing into them + flags --
    protocol, flags, window alternation and bits as on real peers; what one GPU cannot show is visibility across xGMI."""
    peer = True

    def exchange_handles(self, handles):
        return list(handles)

