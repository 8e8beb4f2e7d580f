"""Randomised differential test of the whole create path: random genomes (alphabet, repeats, duplications), read
lengths, error / N rates, k, min_count and correction parameters -> bg_create on the GPU vs the oracle on the CPU,
every container part compared.  Runs until the time budget is used (FUZZ_SCALE=50 multiplies the genome sizes:
fewer, multi-tile / multi-bucket cases).  Usage: python tests/fuzz_parity.py [seconds] [seed]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import biograph_amd as bg
from oracle import oracle as O
from tests import util


def main():
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
    seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    ctx = bg.Context(0)
    t0 = time.time()
    n = int(os.environ.get("FUZZ_FIRST", "0"))
    while time.time() - t0 < budget:
        seed = seed0 * 100000 + n
        rng = np.random.default_rng(seed)
        k = int(rng.choice([14, 15, 18, 22, 25, 30, 31, 12, 9]))
        rl = int(rng.choice([40, 70, 100, 150, 200, 255]))
        if rl < k + 5:
            rl = k + 20
        alphabet = str(rng.choice(["ACGT", "ACGT", "ACGT", "AC", "ACG", "AT"]))
        glen = int(rng.integers(300, 4000)) * int(os.environ.get("FUZZ_SCALE", "1"))
        cov = int(rng.integers(5, 40))
        n_reads = max(20, glen * cov // rl)
        repeat = (int(rng.integers(3, 40)), int(rng.integers(5, 60))) if rng.random() < 0.5 else None
        reads = util.synth_reads(seed, genome_len=glen, n_reads=n_reads, read_len=rl, err=float(rng.choice([0.0, 0.005, 0.02])),
                                 n_rate=float(rng.choice([0.0, 0.002, 0.02])), alphabet=alphabet, var_len=bool(rng.random() < 0.6),
                                 repeat=repeat)
        if rng.random() < 0.3:  # PCR pile-up
            reads += [reads[int(rng.integers(0, len(reads)))]] * int(rng.integers(10, 300))
        if rng.random() < 0.2:
            reads += ["N" * rl, "A" * rl, "ACGT"[: min(4, rl)], ""]
        mc = int(rng.choice([1, 2, 3, 5]))
        maxc = int(rng.choice([0, 2, 8]))
        mgr = int(rng.choice([0, 2, 3]))
        trim = float(rng.choice([0.0, 0.5, 0.7, 0.9]))
        prm = dict(kmer_size=k, min_kmer_count=mc, max_corrections=maxc, min_good_run=mgr, trim_after_portion=trim)
        ss, st = ctx.create(reads, **prm)
        if os.environ.get("FUZZ_SHARDED"):  # the multi-rank build (ranks simulated in this process) vs the single-GPU one
            import torch
            from biograph_amd import sharded
            P = int(rng.integers(1, 7))
            cuts = sorted(int(x) for x in rng.integers(0, len(reads) + 1, P - 1))
            bounds = [0] + cuts + [len(reads)]
            inputs = {}
            for r in range(P):
                sq, of = O.pack_strings(reads[bounds[r]:bounds[r + 1]])
                dsq = torch.from_numpy(sq.copy()).cuda() if len(sq) else torch.zeros(0, dtype=torch.uint8, device="cuda")
                inputs[r] = (dsq, torch.from_numpy(of.astype(np.int64)).cuda(), bounds[r + 1] - bounds[r])
            p2 = ctx.default_params(**prm)
            out = sharded.sharded_create(sharded.LocalSimComm(P), {r: ctx for r in range(P)}, inputs, p2)
            s2 = out[0][0]
            if s2.size() != ss.size() or s2.parts() != ss.parts():
                print("SHARDED MISMATCH seed", seed, prm, "P", P, flush=True)
                return 1
            s2.free()
        if os.environ.get("FUZZ_READMAP"):  # staged path + readmap (paired or not) + merge of two halves
            r = ctx.pack_reads(reads)
            ks = ctx.kmer_count(r, k, mc)
            c, _ = ctx.correct_reads(r, ks, maxc, mgr, trim)
            cseq, coffs = c.download()
            corrected = O.unpack_strings(cseq, coffs)
            s3, _ = ctx.build_seqset(c, k)
            paired = bool(rng.random() < 0.5) and len(reads) % 2 == 0
            rm = ctx.build_readmap(s3, c, paired)
            kept = [x for x in corrected if x]
            sq, of = O.pack_strings(kept)
            ent = O.Entries.build(sq, of, "simple", k)
            oss3 = O.Seqset.from_entries(ent)
            if paired:
                records = [tuple(x for x in corrected[i:i + 2] if x) for i in range(0, len(corrected), 2)]
                records = [t for t in records if t]
            else:
                records = [(x,) for x in kept]
            orm = O.Readmap(oss3, ent, records, paired)
            if rm.size() != len(orm) or any(rm.parts()[nm] != orm.parts()[nm] for nm in bg.READMAP_PART_NAMES):
                print("READMAP MISMATCH seed", seed, prm, "paired", paired, flush=True)
                return 1
            clean = [x for x in kept if len(x) >= 12 and "N" not in x]
            if clean and paired and s3.size() > 4:  # G4: rc range paths + mates of propagate_from_end
                code = {c: i for i, c in enumerate("ACGT")}
                n3 = s3.size()
                seqs, ends, marks = [], [], []
                for _ in range(12):
                    rd = clean[int(rng.integers(0, len(clean)))]
                    ln = int(rng.integers(1, len(rd)))
                    sq_ = np.array([3 - code[c] for c in reversed(rd[:ln])], np.uint8)
                    seqs.append(sq_)
                    ends.append([(0, n3, 0)])
                    marks.append(sorted(set(int(x) for x in rng.integers(1, max(len(sq_), 2), 3) if 0 < x < len(sq_))))
                gotp = s3.propagate_from_end_batch(seqs, ends, rm, 20, marks, mate_cap=8192)
                for kk in range(len(seqs)):
                    want_ent, want_mates = oss3.propagate_from_end(seqs[kk], ends[kk], orm, 20, marks[kk])
                    want_ent.pop(len(seqs[kk]), None) if len(seqs[kk]) else None
                    ent_, mates_ = gotp[kk]
                    if ent_ != {o_: v_ for o_, v_ in want_ent.items() if o_ != len(seqs[kk]) or o_ == 0} or \
                            mates_ != sorted(want_mates, key=lambda r_: (r_[0], -r_[1], r_[2])):
                        print("PROPAGATE MISMATCH seed", seed, prm, "case", kk, flush=True)
                        return 1
            half = len(kept) // 2
            if half >= 2:
                sa, _ = ctx.build_seqset(ctx.pack_reads(kept[:half]), k)
                sb, _ = ctx.build_seqset(ctx.pack_reads(kept[half:]), k)
                merged, maps = ctx.merge_seqsets([sa, sb])
                if merged.size() != s3.size() or merged.parts() != s3.parts():
                    print("MERGE MISMATCH seed", seed, prm, flush=True)
                    return 1
        oss, oent, info = O.build_seqset_from_fastq_reads(reads, k, mc, maxc, mgr, trim)
        if os.environ.get("FUZZ_WALK") and ss.size() == len(oent) and ss.size() > 4:
            # chained push_front / push_front_drop and find on the seqset just built vs the oracle's walk
            nn = ss.size()
            q = min(4000, 20 * nn)
            begin = rng.integers(0, nn, q).astype(np.uint64)  # whole entries: valid (range, context) triples
            end = begin + 1
            sizes_all, _ = ss.arrays()
            size = sizes_all[begin.astype(np.int64)].astype(np.uint32)
            for _round in range(5):
                if not len(begin):
                    break
                base = rng.integers(0, 4, len(begin)).astype(np.uint8)
                minctx = rng.integers(0, rl, len(begin)).astype(np.uint32)
                ob, oe, osz = oss.push_front_drop(begin, end, size, base, minctx)
                gb, ge, gs = ss.push_front_drop_batch(begin, end, size, base, minctx)
                pb, pe, ps = oss.push_front(begin, end, size, base)
                hb, he, hs = ss.push_front_batch(begin, end, size, base)
                if not ((gb == ob).all() and (ge == oe).all() and (gs == osz).all() and (hb == pb).all() and
                        (he == pe).all() and (hs == ps).all()):
                    print("WALK MISMATCH seed", seed, prm, flush=True)
                    return 1
                okm = oe > ob
                begin, end, size = ob[okm], oe[okm], osz[okm]
            strs = oent.strings()
            qs = [strs[int(i)][int(j):int(j) + int(l)] for i, j, l in zip(rng.integers(0, nn, 60), rng.integers(0, 30, 60),
                                                                       rng.integers(1, 50, 60))]
            qs = [x for x in qs if x] + ["ACGT", "T"]
            fb, fe = ss.find_batch(qs)
            for sq_, b_, e_ in zip(qs, fb, fe):
                wb, we, _ = oss.find(sq_)
                if not ((int(b_), int(e_)) == (wb, we) or (b_ >= e_ and wb >= we)):
                    print("FIND MISMATCH seed", seed, prm, sq_, flush=True)
                    return 1
        if os.environ.get("FUZZ_DISCOVER") and ss.size() == len(oent) and ss.size() > 4:
            # frontier kernels (G2 one step, G3 branch trace) vs the oracle, from frontiers reached by walking reads
            code = {c: i for i, c in enumerate("ACGT")}
            nn = ss.size()
            clean = [x for x in reads if len(x) >= 24 and "N" not in x]
            if clean:
                fr = []
                seqs, fronts = [], []
                for _ in range(24):
                    rd = clean[int(rng.integers(0, len(clean)))]
                    arr = np.array([code[c] for c in rd], np.uint8)
                    cut = int(rng.integers(8, len(arr) - 4))
                    cur = (0, nn, 0)
                    for c in arr[:cut]:  # rc context: push the complement of the forward bases
                        b_, e_, s_ = oss.push_front_drop(np.array([cur[0]], np.uint64), np.array([cur[1]], np.uint64),
                                                         np.array([cur[2]], np.uint32), np.array([3 - c], np.uint8),
                                                         np.array([0], np.uint32))
                        cur = (int(b_[0]), int(e_[0]), int(s_[0]))
                        if cur[1] <= cur[0]:
                            break
                    if cur[1] > cur[0]:
                        fr.append([cur])
                        seqs.append(arr[cut:].copy())
                        fronts.append([cur])
                fr += [[(0, nn, 0)], []]
                min_ctx = [int(x) for x in rng.integers(0, 40, len(fr))]
                got = ss.trace_one_base_batch(fr, min_ctx)
                for f_, mc_, g_ in zip(fr, min_ctx, got):
                    best, cands, n2 = oss.trace_one_base(f_, mc_)
                    if not (g_[0] == best and g_[1] == cands and g_[2] == n2):
                        print("TRACE MISMATCH seed", seed, prm, flush=True)
                        return 1
                if seqs:
                    mo = int(rng.integers(5, 40))
                    gotb = ss.branch_trace_batch(seqs, fronts, mo, ev_cap=512)
                    for s_, f_, g_ in zip(seqs, fronts, gotb):
                        nb_, ev_, cur_ = oss.branch_trace(s_, f_, mo)
                        if not (g_[0] == nb_ and g_[1] == ev_ and g_[2] == cur_):
                            print("BRANCH MISMATCH seed", seed, prm, "mo", mo, "front", f_, "len", len(s_), flush=True)
                            print(" gpu   ", g_[0], g_[1][:6], g_[2][:4], flush=True)
                            print(" oracle", nb_, ev_[:6], cur_[:4], flush=True)
                            return 1
                    # the extension trace (push_to_pair's loop) and propagate_from_end from the same one-range frontiers: the
                    # packed kernels of round 6 (sixteen traces a wave / a lane an assembly) on odd alphabets and short reads
                    steps = int(rng.integers(3, 250))  # (the oracle's seen set has no cap: fewer steps than the list's 256 places)
                    gott = ss.push_to_pair_trace_batch(fronts, mo, max_steps=steps)
                    for f_, g_ in zip(fronts, gott):
                        if g_ != oss.push_to_pair_trace(f_, mo, steps):
                            print("P2P MISMATCH seed", seed, prm, "mo", mo, "steps", steps, "front", f_, flush=True)
                            return 1
                    rcs = [(3 - s_[::-1]).astype(np.uint8) for s_ in seqs]
                    gotp = ss.propagate_from_end_batch(rcs, fronts)
                    for s_, f_, g_ in zip(rcs, fronts, gotp):
                        want_ent, _ = oss.propagate_from_end(s_, f_)
                        if g_[0] != {0: want_ent[0]}:
                            print("PROPAGATE MISMATCH seed", seed, prm, "front", f_, "len", len(s_), flush=True)
                            return 1
        ok = ss.size() == len(oent)
        if ok:
            got, want = ss.parts(), oss.parts()
            ok = all(got[name] == want[name] for name in bg.PART_NAMES)
        ok = ok and st["n_kept_kmers"] == info["kept_kmers"] and st["n_corrected_reads"] == info["corrected_reads"]
        if not ok:
            print("MISMATCH seed", seed, prm, "rl", rl, alphabet, glen, cov, repeat, "entries", ss.size(), len(oent), flush=True)
            return 1
        ss.free()
        n += 1
    print("fuzz: %d random builds equal to the oracle in %.0f s (seed %d)" % (n, time.time() - t0, seed0))
    return 0


sys.exit(main())
