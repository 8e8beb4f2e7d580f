"""TEST ONLY: a stand-in for tests.variants_twin.Walker that answers the three batched walks with the
oracle's literal restatements instead of the GPU kernels, so the host drivers of tests/variants_twin.py
can be checked on the CPU against the reference's own known answers."""
import numpy as np

from tests import variants_twin as V
from oracle import oracle as O


class OracleWalker(V.Walker):
    def __init__(self, oss, orm=None):
        self.ss = oss
        self.rm = orm
        self.n = len(oss)
        self._shared = np.asarray(oss.arrays()[1])

    def propagate(self, seqs_rc, ends, marks, with_mates, max_pairs):
        out = []
        for s, e, m in zip(seqs_rc, ends, marks):
            ent, mates = self.ss.propagate_from_end(V._codes(s), e, self.rm if with_mates else None, max_pairs, m)
            ent.pop(len(s), None) if len(s) else None
            out.append((ent, mates))
        return out

    def branch_trace(self, seqs, fronts, min_overlap):
        return [self.ss.branch_trace(V._codes(s), f, min_overlap, cap=max(len(s), 1)) for s, f in zip(seqs, fronts)]

    def push_trace(self, fronts, min_overlap, max_steps):
        return [self.ss.push_to_pair_trace(f, min_overlap, max_steps) if f else (0, [], []) for f in fronts]


def world_for_reads(unpaired, paired=()):
    """seqset_for_reads + readmap_for_reads of the reference's test utilities (bio_base/seqset_testutil.cpp):
    the reads as given (no correction)."""
    all_reads = list(unpaired) + [r for p in paired for r in p]
    seq, offs = O.pack_strings(all_reads)
    ent = O.Entries.build(seq, offs, "simple")
    oss = O.Seqset.from_entries(ent)
    records = [(r,) for r in unpaired] + [tuple(p) for p in paired]
    orm = O.Readmap(oss, ent, records, True)
    return oss, orm, ent
