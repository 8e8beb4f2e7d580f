/* bgseq.h — C ABI of libbgseq.so: MI355X-native seqset construction and overlap walk.
 *
 * The reference (spiralgenetics/biograph) exposes no FFI for this path: it is linked C++
 * (SURVEY.md §8b).  Each entry point below names the reference seam (file:line under the
 * reference tree) whose work it replaces; INTEGRATION.md shows the C++ façade a maintainer
 * would put behind `SEQSETMain::run` / `seqset_range` to call it.
 *
 * Conventions
 *   - plain pointers and sizes only; `d_` parameters are DEVICE pointers (hipMalloc'ed or the
 *     data_ptr of a torch tensor on the same device), `h_` parameters are HOST pointers;
 *   - every call returns 0 on success or a negative BG_E* code; bg_last_error() returns a
 *     thread-local message for the last failure on the calling thread;
 *   - all device work of a context is issued on the HIP stream given at bg_ctx_create() and the
 *     call returns after that work has completed unless stated otherwise;
 *   - there is NO CPU fallback: without a usable HIP device every compute entry point fails
 *     with BG_ENODEV.
 */
#ifndef BGSEQ_H_
#define BGSEQ_H_

#include <stddef.h>
#include <stdint.h>
#include <sys/types.h>

#ifdef __cplusplus
extern "C" {
#endif

#define BG_OK 0
#define BG_EINVAL (-1)   /* bad argument (reference: io_exception / CHECK failure)            */
#define BG_ENODEV (-2)   /* no HIP device / HIP runtime error                                  */
#define BG_ENOMEM (-3)   /* device or host allocation failed                                   */
#define BG_EBADBASE (-4) /* read contains a character outside ACGTN (dna_base.h:54 throws)     */
#define BG_EIO (-5)      /* file I/O error                                                     */
#define BG_EINTERNAL (-6)/* invariant violated (reference: "Missing expansion?" builder.cpp:96)*/

#define BG_ABI_VERSION 2

typedef struct bg_ctx bg_ctx;         /* device context: stream + workspace cache              */
typedef struct bg_reads bg_reads;     /* 2-bit packed reads resident in HBM                    */
typedef struct bg_kmerset bg_kmerset; /* filtered canonical k-mer set resident in HBM          */
typedef struct bg_seqset bg_seqset;   /* seqset arrays resident in HBM (+ host copies on demand)*/
typedef struct bg_records bg_records; /* suffix records (materialised keys) of one shard           */

int bg_abi_version(void);
int bg_device_count(void); /* HIP devices this process sees (0 without a usable runtime); initialises HIP: not before a fork */
const char* bg_last_error(void);

/* stream: a hipStream_t (e.g. torch.cuda.current_stream().cuda_stream) or NULL for the default
 * stream. */
int bg_ctx_create(int device, void* stream, bg_ctx** out);
/* The same with a (non-blocking) stream of its own that lives as long as the context: for callers that run several
 * contexts side by side from several host threads (bgbinary discovery's tracer sessions). */
int bg_ctx_create_own_stream(int device, bg_ctx** out);
/* Pre-faults `bytes` of device memory into the context's allocator (the driver maps memory at ~30 us per MiB the first time;
 * a one-shot command can pay that on a side thread while it reads its input).  May be called from another thread than the
 * one using the context. */
int bg_ctx_reserve(bg_ctx* ctx, uint64_t bytes);
/* Every object created from a context (reads, k-mer sets, records, seqsets) must be freed BEFORE the
 * context is destroyed. */
void bg_ctx_destroy(bg_ctx* ctx);
/* Releases cached workspace blocks back to the driver. */
int bg_ctx_trim(bg_ctx* ctx);
/* Device memory of the context's allocator, bytes: in use now, cached for reuse, high-water mark of the
 * in-use figure (reset = 1 restarts the mark from the current figure).  Any pointer may be NULL. */
int bg_ctx_memory(bg_ctx* ctx, uint64_t* live, uint64_t* cached, uint64_t* peak, int reset_peak);

/* Per-stage device time of the most recent build call on this context (HIP events on the
 * context's stream).  Names follow runtime_stats stages of biograph_create.cpp:540-783 plus the
 * kernel families.  Fills up to cap entries; returns the number available. */
typedef struct bg_timing {
  char name[48];
  double ms;        /* summed device time                                                      */
  uint64_t launches;/* kernel launches accounted                                               */
  uint64_t bytes;   /* algorithmic bytes moved by those launches, 0 if n/a  */
} bg_timing;
int bg_ctx_timings(bg_ctx* ctx, bg_timing* out, int cap);
/* Enable (1) / disable (0) per-kernel-family event timing (off by default; costs event pairs). */
int bg_ctx_set_profiling(bg_ctx* ctx, int on);

/* ---- S1: import.  Replaces read_importer + dna_sequence packing
 * (modules/build_seqset/read_importer.h:65-141, modules/bio_base/dna_sequence.h:97).
 * d_seq: concatenated read characters (ACGTN, either case for ACGT); d_offs: n_reads+1 offsets. */
int bg_pack_reads(bg_ctx* ctx, const uint8_t* d_seq, const uint64_t* d_offs, uint64_t n_reads,
                  bg_reads** out);
/* Convenience for host callers (CLI, tests): uploads then packs. */
int bg_pack_reads_host(bg_ctx* ctx, const uint8_t* h_seq, const uint64_t* h_offs, uint64_t n_reads,
                       bg_reads** out);
uint64_t bg_reads_count(const bg_reads* r);
uint64_t bg_reads_total_bases(const bg_reads* r); /* sum of lengths (0-length = dropped read)   */
uint32_t bg_reads_max_len(const bg_reads* r);
/* Unpacks to ASCII on the host: h_seq must hold bg_reads_total_bases() bytes, h_offs n+1. */
int bg_reads_download(bg_ctx* ctx, const bg_reads* r, uint8_t* h_seq, uint64_t* h_offs);
void bg_reads_free(bg_reads* r);

/* FASTQ import on the device (csrc/fastq.hip).  Replaces fastq_reader + read_importer_base::read_fastq
 * (modules/build_seqset/read_importer.cpp:604, read_importer.h:65-141): d_text[s] = the bytes of source s as
 * they are in the file (plain text; gzip is inflated by the caller), resident in HBM.  Four lines per record,
 * '@' / '+' lines checked, qualities dropped (read_importer.cpp:22), "\r\n" accepted, blank lines at the end
 * ignored.  paired_files: sources 2i and 2i+1 are mate files read in lock step; interleaved: records 2i, 2i+1
 * of a source are mates; either way mates are reads 2i, 2i+1 of the result.  cut_end != 0: bases
 * [cut_start, min(cut_end, len)) of every read (--cut-reads, read_importer.cpp:157-169).  The results
 * (*d_seq_out: concatenated bases, *d_offs_out: n+1 offsets — what bg_pack_reads takes) are allocated from
 * the context: release them with bg_dev_free. */
int bg_fastq_import(bg_ctx* ctx, uint32_t n_src, const uint8_t* const* d_text, const uint64_t* h_bytes, int paired_files,
                    int interleaved, uint32_t cut_start, uint32_t cut_end, uint8_t** d_seq_out, uint64_t** d_offs_out,
                    uint64_t* n_reads_out, uint64_t* n_bases_out);
/* Host -> device copy through two pinned staging buffers filled by `threads` host threads (0 = all cores)
 * while the DMA engine drains the other one. */
int bg_dev_upload_pinned(bg_ctx* ctx, void* d_dst, const void* h_src, uint64_t bytes, int threads);
/* the same from an open file descriptor: `bytes` bytes from file_offset, pread() by the filling threads */
int bg_dev_upload_file(bg_ctx* ctx, void* d_dst, int fd, uint64_t file_offset, uint64_t bytes, int threads);

/* ---- S2: k-mer counting + filter.  Replaces kmer_counter + kmerizer::run + kmer_set
 * (modules/build_seqset/kmer_counter.h:123-326, modules/bio_mapred/kmerize_bf.cpp:267-429).
 * Keeps canonical k-mers with fwd+rev >= min_count; flags bit0 = fwd_starts_read,
 * bit1 = rev_starts_read (kmer_set.h:20-21).  1 <= k <= 31. */
int bg_kmer_count(bg_ctx* ctx, const bg_reads* reads, int k, uint32_t min_count, bg_kmerset** out);
uint64_t bg_kmerset_size(const bg_kmerset* ks);
uint64_t bg_kmerset_distinct(const bg_kmerset* ks);  /* distinct canonical k-mers before filter */
uint64_t bg_kmerset_instances(const bg_kmerset* ks); /* k-mer instances counted (N_kmers)       */
/* Sorted canonical k-mers, their flags and total counts, to host arrays of bg_kmerset_size(). */
int bg_kmerset_download(bg_ctx* ctx, const bg_kmerset* ks, uint64_t* h_kmers, uint8_t* h_flags,
                        uint32_t* h_counts);
void bg_kmerset_free(bg_kmerset* ks);

/* ---- S3: read correction.  Replaces correct_reads::correct + fast_read_correct
 * (modules/build_seqset/correct_reads.cpp:154-231, modules/bio_base/fast_read_correct.cpp:16-182).
 * Dropped reads come back with length 0.  max_corrections <= 8. */
typedef struct bg_correct_params {
  uint32_t max_corrections;   /* --max-corrections, default 8   (biograph_create.cpp:288)      */
  uint32_t min_good_run;      /* --min-good-run, default 2                                      */
  float trim_after_portion;   /* --trim-after-portion, default 0.7f (parsed as float, :489)     */
} bg_correct_params;
typedef struct bg_correct_stats {
  uint64_t corrected_reads, corrected_bases, failed, modified, modified_bases, truncated,
      dropped_bases;
} bg_correct_stats;
int bg_correct_reads(bg_ctx* ctx, const bg_reads* reads, const bg_kmerset* ks,
                     const bg_correct_params* params, bg_reads** out, bg_correct_stats* stats);

/* ---- S4+S5: seqset from corrected reads.  Replaces part_repo/expander/builder
 * (modules/build_seqset/expand.cpp:418,581; builder.cpp:8,207; bio_base/seqset.cpp:19-44,113-129).
 * Result = prefix-free maxima of all suffixes of all reads and their reverse complements
 * (SURVEY row E), with sizes/shared/prev_{ACGT} rank vectors/fixed. */
typedef struct bg_build_stats {
  uint64_t n_reads, n_bases;         /* input                                                  */
  uint64_t n_kmers, n_distinct_kmers, n_kept_kmers;
  uint64_t n_corrected_reads, n_corrected_bases;
  uint64_t n_records;                /* suffix records sorted (N_rec)                          */
  uint64_t n_tied_records;           /* records in first-word tie groups > 32 (radix-refined)  */
  uint64_t n_entries;                /* N_ent                                                  */
  uint32_t max_len;
} bg_build_stats;
int bg_seqset_build(bg_ctx* ctx, const bg_reads* corrected, int cover_k, bg_seqset** out,
                    bg_build_stats* stats);

/* Whole `biograph create` data path on device-resident input: S1..S5
 * (SEQSETMain::run, modules/biograph/biograph_create.cpp:431-783 minus readmap/metadata). */
typedef struct bg_create_params {
  int kmer_size;               /* --kmer-size 30        */
  uint32_t min_kmer_count;     /* --min-kmer-count 5    */
  bg_correct_params correct;
} bg_create_params;
void bg_create_params_default(bg_create_params* p);
int bg_create(bg_ctx* ctx, const uint8_t* d_seq, const uint64_t* d_offs, uint64_t n_reads,
              const bg_create_params* p, bg_seqset** out, bg_build_stats* stats);

/* ---- seqset object (modules/bio_base/seqset.h:35-128). */
uint64_t bg_seqset_size(const bg_seqset* ss);
uint32_t bg_seqset_max_len(const bg_seqset* ss);
/* seqset::uuid() (seqset.h:35-128; file_info.json of the container): NUL-terminated into h_out[cap]. */
int bg_seqset_uuid(const bg_seqset* ss, char* h_out, uint64_t cap);
/* Host copies of the payload parts exactly as stored in the spiral container (format 1.1.0):
 * part in {"fixed","entry_sizes/elements","shared/elements","prev_A/bits","prev_A/subaccum",
 * "prev_A/accum", ... "prev_T/accum"}.  Call with h_out=NULL to get the byte size in *size. */
int bg_seqset_part(bg_ctx* ctx, const bg_seqset* ss, const char* part, void* h_out, uint64_t* size);
/* Unpacked arrays (uint32 per entry) for tests / SDK: sizes and shared. */
int bg_seqset_arrays(bg_ctx* ctx, const bg_seqset* ss, uint32_t* h_sizes, uint32_t* h_shared);
/* Sequences of entries [first, first+count) as ASCII (seqset_range::sequence, seqset.cpp:609). */
int bg_seqset_entries(bg_ctx* ctx, const bg_seqset* ss, uint64_t first, uint64_t count,
                      uint8_t* h_seq, uint64_t* h_offs);
/* Spiral zip container (modules/io/spiral_file_mmap.cpp:350-457, spiral_file.cpp:194-222).
 * argv/argc go to file_info.json "command_line". */
int bg_seqset_write(bg_ctx* ctx, const bg_seqset* ss, const char* path, int argc,
                    const char* const* argv);
int bg_seqset_open(bg_ctx* ctx, const char* path, bg_seqset** out);
/* Upload from host arrays (tests, goldens in the old raw-uint8 layout). bits: 4 arrays of
 * ceil(n/64) words. */
int bg_seqset_from_arrays(bg_ctx* ctx, uint64_t n, const uint32_t* h_sizes, const uint32_t* h_shared,
                          const uint64_t* const h_bits[4], bg_seqset** out);
void bg_seqset_free(bg_seqset* ss);

/* ---- G1: batched walk primitives over a resident seqset.  Replace seqset_range::push_front
 * (seqset.cpp:319-338), push_front_drop (:350-429) and seqset::find (:146-155).
 * All arrays are DEVICE pointers of n elements; base in 0..3 (A,C,G,T).  Invalid result ranges
 * are (0,0,0) for push_front_drop and begin>=end for push_front, as in the reference. */
int bg_push_front_batch(bg_ctx* ctx, const bg_seqset* ss, uint64_t n, const uint64_t* d_begin,
                        const uint64_t* d_end, const uint32_t* d_size, const uint8_t* d_base,
                        uint64_t* d_obegin, uint64_t* d_oend, uint32_t* d_osize);
int bg_push_front_drop_batch(bg_ctx* ctx, const bg_seqset* ss, uint64_t n, const uint64_t* d_begin,
                             const uint64_t* d_end, const uint32_t* d_size, const uint8_t* d_base,
                             const uint32_t* d_min_ctx, uint64_t* d_obegin, uint64_t* d_oend,
                             uint32_t* d_osize);
/* seqset_range::pop_front (bio_base/seqset.cpp:609-628; the walk primitive of the pop tracer, variants/pop_tracer.cpp):
 * the range of the sequence without its first base.  An invalid or empty input range (the reference throws) gives
 * (0, 0, 0). */
int bg_pop_front_batch(bg_ctx* ctx, const bg_seqset* ss, uint64_t n, const uint64_t* d_begin, const uint64_t* d_end,
                       const uint32_t* d_size, uint64_t* d_obegin, uint64_t* d_oend, uint32_t* d_osize);
/* find: queries are concatenated ACGT strings (device), n+1 offsets; result ranges (begin,end). */
int bg_find_batch(bg_ctx* ctx, const bg_seqset* ss, uint64_t n, const uint8_t* d_seq,
                  const uint64_t* d_offs, uint64_t* d_obegin, uint64_t* d_oend);

/* Measurement only: counts the push_front_drop calls and the iterations of its drop loop (seqset.cpp:368-421) made by the walk
 * kernels of this process between on = 1 and on = 0 (h_out2[0] = calls, [1] = drop iterations): N_pfd and N_drop of the
 * walk's algorithmic bytes, 52 N_pfd + 60 N_drop.  Two contended atomics: not for timed runs. */
int bg_walk_counters(bg_ctx* ctx, int on, uint64_t* h_out2 /* may be NULL */);

/* ---- G2/G3/G7: frontier walks.  A frontier is a set of at most 16 ranges; frontier f owns ranges
 * [d_foff[f], d_foff[f+1]) of (d_begin, d_end, d_size).  Output frontiers use slots f*16 .. f*16+15. */
/* One step of push_to_pair_discover::trace_one_base (graph_discover/push_to_pair.cpp:159-240) for every
 * frontier: all 4 bases x ranges push_front_drop(rc_b, d_min_ctx[f]); the base with the longest result
 * wins (first base on ties); d_best_base = that rc base or 255; candidates = results of that base
 * before prefix-dedup with their "dropped" flag (the host keeps m_seen_ranges, :220-224); next = the
 * returned frontier after seqset_set_dedup_prefixes (variants/assemble.cpp:1473-1498), in set order. */
int bg_trace_one_base_batch(bg_ctx* ctx, const bg_seqset* ss, uint64_t n_frontiers, const uint32_t* d_foff,
                            const uint64_t* d_begin, const uint64_t* d_end, const uint32_t* d_size,
                            const uint32_t* d_min_ctx, uint8_t* d_best_base, uint8_t* d_n_cand,
                            uint64_t* d_cbegin, uint64_t* d_cend, uint32_t* d_csize, uint8_t* d_cdrop,
                            uint8_t* d_n_next, uint64_t* d_nbegin, uint64_t* d_nend, uint32_t* d_nsize);
/* branch_discover::on_trace (graph_discover/branch.cpp:20-80) for n_asm assemblies: assembly a walks bases
 * d_seq_codes[d_seq_off[a] .. d_seq_off[a+1]) (0..3) from its frontier; events = (offset, alternative base)
 * of every 1-base branch (first ev_cap stored per assembly, d_n_ev = total); out = final frontier. */
int bg_branch_trace_batch(bg_ctx* ctx, const bg_seqset* ss, uint64_t n_asm, const uint64_t* d_seq_off,
                          const uint8_t* d_seq_codes, const uint32_t* d_foff, const uint64_t* d_begin,
                          const uint64_t* d_end, const uint32_t* d_size, uint32_t min_overlap, uint32_t ev_cap,
                          uint32_t* d_ev_off, uint8_t* d_ev_base, uint32_t* d_n_ev, uint8_t* d_n_out,
                          uint64_t* d_obegin, uint64_t* d_oend, uint32_t* d_osize);
/* The extension trace of push_to_pair_discover::on_trace (graph_discover/push_to_pair.cpp:271-343) for
 * n_asm assemblies at once: from the assembly's rc start ranges (d_foff / d_begin / d_end / d_size as in
 * bg_trace_one_base_batch, min context = min_overlap: k_check_reads is false, :14,203-232) repeat
 * trace_one_base until it returns nothing.  Per assembly a: d_n_steps[a] steps were taken; step t
 * appended base d_bases[a*max_steps+t] and left a frontier of d_counts[a*max_steps+t] ranges, stored
 * consecutively in d_rbegin/d_rend/d_rsize[a*range_cap ..] (d_n_ranges[a] in total).  d_reason[a]:
 * 0 nothing extends, 1 a dropped range repeated (m_seen_ranges, :221-225), 2 max_steps reached,
 * 3 more than seen_cap dropped ranges, 4 range buffer full (resume from the last frontier for 2-4).
 * The mate / anchor checks between the steps (:289-342) stay with the host: they cut the path, they do
 * not change it. */
int bg_push_to_pair_trace_batch(bg_ctx* ctx, const bg_seqset* ss, uint64_t n_asm, const uint32_t* d_foff,
                                const uint64_t* d_begin, const uint64_t* d_end, const uint32_t* d_size,
                                uint32_t min_overlap, uint32_t max_steps, uint32_t range_cap, uint32_t seen_cap,
                                uint32_t* d_n_steps, uint8_t* d_reason, uint8_t* d_bases, uint8_t* d_counts,
                                uint32_t* d_n_ranges, uint64_t* d_rbegin, uint64_t* d_rend, uint32_t* d_rsize);
/* ref_map::build (variants/ref_map.cpp:72-185): marks seqset entries that are full-entry matches of the
 * reference (both strands).  d_ref_codes = reference bases 0..3; extents = N-free supercontigs
 * (flat_ref index, bio_base/flat_ref.h).  d_out[n] = [fwd_flag:1|rev_flag:1|count:6 saturating]
 * (ref_map.h:49-53).  chain_bases = bases per GPU chain (0 = chosen from the reference's size so that enough chains are in flight: 1024..4096;
 * the result does not depend on it). */
int bg_ref_map_build(bg_ctx* ctx, const bg_seqset* ss, const uint8_t* d_ref_codes, uint64_t n_extents,
                     const uint64_t* h_ext_start, const uint64_t* h_ext_size, uint32_t chain_bases,
                     uint8_t* d_out);

/* The reference map as a file (`biograph discovery --ref-map`, biograph_variants.cpp:171,643-666): a spiral container with one
 * packed_vector<unsigned, 8>, one byte per seqset entry as bg_ref_map_build leaves them (ref_map.cpp:24-38).  HOST arrays.
 * bg_ref_map_read fails when the file holds another number of entries (ref_map.cpp:37). */
int bg_ref_map_write(const char* path, const uint8_t* h_map, uint64_t n, int argc, const char* const* argv);
int bg_ref_map_read(const char* path, uint8_t* h_map_out, uint64_t n);

/* ---- Readmap: make_readmap::create_from_reads
 * (modules/bio_mapred/make_readmap.cpp:120-392) over the corrected reads of one sample.
 * `ss` must hold resident entry keys (a seqset built in this process by bg_seqset_build / bg_create);
 * `corrected` is what bg_correct_reads returned (dropped reads have length 0).  With is_paired != 0,
 * reads 2i and 2i+1 are mates (the order bg_pack_reads received them in; unpaired_reads / paired_reads
 * of corrected_reads.h); a pair whose mate was dropped contributes the surviving read unpaired
 * (biograph_create.cpp:880-889).  Read ids are the rows of the (entry, type, lengths, loop entry)
 * total order (make_readmap.h:187-213).  BG_EINTERNAL if a read is absent from the seqset (:147-151). */
typedef struct bg_readmap bg_readmap;
int bg_readmap_build(bg_ctx* ctx, const bg_seqset* ss, const bg_reads* corrected, int is_paired,
                     bg_readmap** out);
/* The same on a MERGED seqset for one of its input samples: d_marks (one byte per entry, from
 * bg_seqset_mark_inputs; NULL = bg_readmap_build) marks the images of the sample's entries under the merge map, and
 * every read goes to the first marked entry that has it as a prefix -- the row make_readmap::create_from_fast_migrate
 * (bio_mapred/make_readmap.cpp:455-519) gives it.  ss may have been opened from a file (FM search instead of keys). */
int bg_readmap_build_marked(bg_ctx* ctx, const bg_seqset* ss, const bg_reads* corrected, int is_paired,
                            const uint8_t* d_marks, bg_readmap** out);
uint64_t bg_readmap_size(const bg_readmap* rm);         /* readmap::size(), number of read ids */
uint32_t bg_readmap_max_read_len(const bg_readmap* rm); /* readmap::max_read_len() */
/* 0 for a legacy readmap that stores mates as a pair table (readmap.cpp:48-51): bg_readmap_arrays then returns the pair
 * pointer (0xffffffff = no mate) in place of the loop pointer, and whatever follows loops refuses the readmap. */
int bg_readmap_has_mate_loop(const bg_readmap* rm);
/* Raw bytes of one mmap part of the container, named as in the spiral file:
 * "read_ids/{source_to_mid,dest_to_mid}/{bits,subaccum,accum}", "read_lengths/elements",
 * "mate_loop_ptr/elements", "is_forward/packed_data".  Same size protocol as bg_seqset_part. */
int bg_readmap_part(bg_ctx* ctx, const bg_readmap* rm, const char* part, void* h_out, uint64_t* size);
/* Writes the readmap container (readmap 1.2.0: sparse_multi read_ids, packed_varbit_vector
 * read_lengths / mate_loop_ptr, 1-bit is_forward; readmap.json binds it to the seqset uuid).
 * h_sha1_hex (41 bytes, may be null) receives the SHA-1 of the file, which names it
 * coverage/<sha1>.readmap in a .bg directory (biograph_create.cpp:827-828). */
int bg_readmap_write(bg_ctx* ctx, const bg_readmap* rm, const char* path, int argc, const char* const* argv,
                     char* h_sha1_hex);
/* The digest bg_readmap_write reports, for any file (sha1_hash over the file, biograph_create.cpp:827-828); uses the x86
 * SHA extensions when the host has them.  Host only. */
int bg_sha1_file(const char* path, char* hex41);
void bg_readmap_free(bg_readmap* rm);
/* A readmap from its per-read columns, all DEVICE arrays of n_reads elements in read-id order: seqset entry (ascending;
 * 64-bit -- a whole-human seqset has > 2^32 entries), read length, mate loop pointer, orientation.  What the ranks of a
 * sharded build gather on the rank that writes the container (biograph_amd/sharded.py sharded_readmap); the
 * sparse_multi bit vectors and their bitcounts are derived here (io/sparse_multi.cpp, make_readmap.cpp:229-368). */
int bg_readmap_from_arrays(bg_ctx* ctx, uint64_t n_entries, uint32_t max_read_len, const char* seqset_uuid, uint64_t n_reads,
                           const uint64_t* d_entry_id, const uint32_t* d_read_len, const uint32_t* d_mate_loop_ptr,
                           const uint8_t* d_is_forward, bg_readmap** out);
/* ---- Sharded readmap of one sample (csrc/readmap.hip; biograph_amd/sharded.py sharded_readmap): the entries are
 * range-partitioned over the ranks (shard q = global entries [h_entry_offsets[q], h_entry_offsets[q+1]) with their keys
 * resident), every rank holds some of the sample's pairs; make_readmap::create_from_reads (make_readmap.cpp:120-392) with
 * one owner per structure.  Rows of u64 words on the device, routed by the host (RCCL all_to_all):
 *   bg_seqset_find_rows            owner of a key range: global entry id of every read row it received (rows as
 *                                  bg_oriented_rows makes them, routed with BG_PART_KEY_BOUNDS; ~0 for an empty row) --
 *                                  seqset::find_existing_unique (seqset.cpp:173-188)
 *   bg_readmap_rows                reader: the mate-loop rows (make_readmap.h:56-90) of its pairs from those entry ids,
 *                                  3 words each: hi = entry:37|type:2|len:10|mate_len:10, lo = loop entry, destination
 *                                  (owner of the row's entry) << 10: route with BG_PART_ROW_DEST
 *   bg_readmap_shard_build         owner: sorts the rows it received (make_readmap.h:187-213); with the ranks in order
 *                                  read id = rows on the ranks before (read_id_base) + position
 *   bg_readmap_shard_first_claims / _apply_claims / _take_replies
 *                                  the three claim hops of make_readmap.cpp:283-392 as 4-word claims (group key, read id
 *                                  of the loop's start, read id of the claiming row, destination << 10): the owner of
 *                                  the group gives its k-th claimer (in read-id order of the loop starts) the group's
 *                                  k-th row, replies with that row's read id (one u64 per claim, in received order) and
 *                                  writes the next hop's claims; the claimer's owner stores the reply as the claiming
 *                                  row's loop pointer
 *   bg_readmap_shard_columns       the shard's stretch of the per-read columns (global entry ids and read ids) for
 *                                  bg_readmap_from_arrays */
typedef struct bg_readmap_shard bg_readmap_shard;
int bg_seqset_find_rows(bg_ctx* ctx, const bg_seqset* shard, const uint64_t* d_rows, uint64_t n, int W, uint64_t entry_base,
                        uint64_t* d_entry);
int bg_readmap_rows(bg_ctx* ctx, const bg_reads* reads, int is_paired, const uint64_t* d_entry /* 2 per read: forward, rc */,
                    uint32_t nparts, const uint64_t* h_entry_offsets /* nparts + 1 */, uint64_t* d_rows /* cap x 3 */,
                    uint64_t cap, uint64_t* n_out);
int bg_readmap_shard_build(bg_ctx* ctx, const uint64_t* d_rows /* n x 3 */, uint64_t n, bg_readmap_shard** out);
uint64_t bg_readmap_shard_size(const bg_readmap_shard* s);
void bg_readmap_shard_free(bg_readmap_shard* s);
int bg_readmap_shard_first_claims(bg_ctx* ctx, const bg_readmap_shard* s, uint64_t read_id_base, uint32_t nparts,
                                  const uint64_t* h_entry_offsets, uint64_t* d_claims /* cap x 4 */, uint64_t cap, uint64_t* n_out);
int bg_readmap_shard_apply_claims(bg_ctx* ctx, bg_readmap_shard* s, uint64_t read_id_base, int hop /* 1..3 */,
                                  const uint64_t* d_claims /* n x 4 */, uint64_t n, uint32_t nparts, const uint64_t* h_entry_offsets,
                                  uint64_t* d_reply /* n */, uint64_t* d_next /* n x 4 */, uint64_t* n_next);
int bg_readmap_shard_take_replies(bg_ctx* ctx, bg_readmap_shard* s, uint64_t read_id_base, const uint64_t* d_claims_sent,
                                  const uint64_t* d_reply, uint64_t n);
int bg_readmap_shard_columns(bg_ctx* ctx, const bg_readmap_shard* s, uint64_t* d_entry_id, uint32_t* d_read_len,
                             uint32_t* d_mate_loop_ptr, uint8_t* d_is_forward);
/* Opens a readmap container (format 1.2.0 with the mate loop table) and keeps it resident. */
int bg_readmap_open(bg_ctx* ctx, const char* path, bg_readmap** out);
/* Whole-array form of the per-read accessors (readmap.h:120-138): seqset entry of the read id
 * (sparse_multi::reverse_lookup), get_readlength, the mate loop pointer, get_is_forward.  Host buffers of
 * bg_readmap_size() elements; any may be null.  Entry ids are 64-bit (a whole-human seqset has > 2^32 entries), read
 * ids 32-bit as in the reference. */
int bg_readmap_arrays(bg_ctx* ctx, const bg_readmap* rm, uint64_t* h_entry_id, uint32_t* h_read_len,
                      uint32_t* h_mate_loop_ptr, uint8_t* h_is_forward);
/* readmap::get_prefix_reads (bio_base/readmap.cpp:434-447, read_iterator :538-649) for n ranges: the ids
 * of the reads whose sequence is a prefix of the range's sequence, in iteration order (upward from the
 * range's first entry, then downward).  d_ids[q*cap ..] receives up to cap ids, d_count[q] the full
 * number.  BG_EINVAL if the readmap belongs to another seqset (:436-438). */
int bg_prefix_reads_batch(bg_ctx* ctx, const bg_seqset* ss, const bg_readmap* rm, uint64_t n,
                          const uint64_t* d_begin, const uint64_t* d_end, const uint32_t* d_size,
                          int read_len_limit, uint32_t cap, uint32_t* d_ids, uint32_t* d_count);
/* readmap::read accessors for n read ids: has_mate (readmap.cpp:207-218), get_mate_rc (:270-290; ~0 when
 * unmated), get_rev_comp (:248-268), get_seqset_entry() = ctx_entry(id).truncate(length)
 * (readmap.h:249-251, seqset.cpp:630-661). */
int bg_read_info_batch(bg_ctx* ctx, const bg_seqset* ss, const bg_readmap* rm, uint64_t n,
                       const uint32_t* d_read_ids, uint8_t* d_has_mate, uint32_t* d_mate_rc, uint32_t* d_rev_comp,
                       uint64_t* d_begin, uint64_t* d_end, uint32_t* d_size);
/* seqset_path::propagate_from_end (variants/assemble.cpp:1345-1418) for n_asm assemblies at once — the
 * kernel of update_rc_seqset_entries::propagate (graph_discover/update_rc_seqset_entries.cpp:66-73).
 * Assembly a pushes d_seq_codes[d_seq_off[a] .. d_seq_off[a+1]) from its LAST base to its first (the
 * caller passes the `seq` argument, i.e. the reverse complement of the assembly) starting from the range
 * set new_ends = ranges d_foff[a] .. d_foff[a+1] (<= 16).  Outputs: the set at offset 0 (starts) in
 * d_obegin/d_oend/d_osize[a*16 ..] with d_n_out[a]; the sets at the marked offsets (d_marks ascending
 * inside d_mark_off[a] .. d_mark_off[a+1], strictly between 0 and the length; d_mark_off may be null)
 * in d_mbegin/...[mark*16 ..] with d_mark_n[mark]; and, when rm != NULL, the seqset entries of
 * get_mate_rc() of the mated prefix reads of every range that has at most max_pairs_per_read of them
 * (:1394-1414): up to mate_cap ranges per assembly, unsorted and with repeats, d_n_mates[a] = full
 * number (the reference keeps them in a set — fold on the host). */
int bg_propagate_from_end_batch(bg_ctx* ctx, const bg_seqset* ss, const bg_readmap* rm, uint64_t n_asm,
                                const uint64_t* d_seq_off, const uint8_t* d_seq_codes, const uint32_t* d_foff,
                                const uint64_t* d_begin, const uint64_t* d_end, const uint32_t* d_size,
                                uint32_t max_pairs_per_read, const uint32_t* d_mark_off, const uint32_t* d_marks,
                                uint8_t* d_mark_n, uint64_t* d_mbegin, uint64_t* d_mend, uint32_t* d_msize,
                                uint8_t* d_n_out, uint64_t* d_obegin, uint64_t* d_oend, uint32_t* d_osize,
                                uint32_t mate_cap, uint32_t* d_n_mates, uint64_t* d_mate_begin,
                                uint64_t* d_mate_end, uint32_t* d_mate_size);
/* ---- Graph-discover host drivers (C++ in csrc/discover.cpp over the batched walks
 * above).  An assembly list mirrors the stream of variants::assembly the reference's pipeline steps pass
 * along (variants/assemble.h:285-420: left_offset / right_offset with INT64_MIN = optional_aoffset::none,
 * seq, matches_reference, tags, rc_seqset_entries).  Every step takes the list in canonical order
 * (canon_assembly_order, assemble.cpp:1029-1085) and leaves it so.  Sequences are upper-case ACGT. */
typedef struct bg_asm_list bg_asm_list;
#define BG_OFFSET_NONE INT64_MIN
int bg_asm_list_create(bg_asm_list** out);
void bg_asm_list_free(bg_asm_list* l);
uint64_t bg_asm_list_size(const bg_asm_list* l);
int bg_asm_list_add(bg_asm_list* l, int64_t left_offset, int64_t right_offset, const char* seq, uint64_t seq_len,
                    int matches_reference, const char* tags_csv);
int bg_asm_list_sort(bg_asm_list* l);
/* Pointers stay valid until the list changes (tags_csv until the next bg_asm_list_get). */
int bg_asm_list_get(bg_asm_list* l, uint64_t i, int64_t* left_offset, int64_t* right_offset, const char** seq,
                    uint64_t* seq_len, int* matches_reference, const char** tags_csv);
/* rc_seqset_entries of assembly i (seqset_path::entries()): offsets ascending, a count and that many
 * (begin, end, size) triples per offset.  Call with null arrays to size them. */
int bg_asm_list_rc_path(const bg_asm_list* l, uint64_t i, uint64_t* n_offsets, uint64_t* n_ranges, int64_t* h_offsets,
                        uint32_t* h_counts, uint64_t* h_triples);
/* make_ref_assemblies (graph_discover/make_ref.cpp:9-35) over scaffold extents; appends. */
int bg_make_ref_assemblies(bg_asm_list* l, uint64_t n_extents, const int64_t* ext_offset, const char* const* ext_seq,
                           const uint64_t* ext_len, int64_t start_offset, int64_t end_offset, int64_t max_chunk_size);
/* graph_trim_ref (graph_discover/graph_trim_ref.cpp): trims reference-matching ends off variant assemblies
 * and cuts reference assemblies where they attach.  max_backtrack 0 = 300 (graph_trim_ref.h:31). */
int bg_graph_trim_ref(bg_asm_list* l, uint64_t n_extents, const int64_t* ext_offset, const char* const* ext_seq,
                      const uint64_t* ext_len, int64_t max_backtrack);
/* The dedup of python/biograph/variants/graph_discover_test.py:23-35. */
int bg_dedup_assemblies(bg_asm_list* l);
/* update_rc_seqset_entries (graph_discover/update_rc_seqset_entries.cpp) on apply_edges_step
 * (variants/apply_edges.cpp); rm != NULL also collects mates (assemble.cpp:1394-1414). */
int bg_update_rc_seqset_entries(bg_ctx* ctx, const bg_seqset* ss, const bg_readmap* rm, bg_asm_list* l,
                                uint32_t max_pairs_per_read);
/* discover_branch / discover_push_to_pair (python/biograph/variants/graph_discover.cpp:110-122;
 * graph_discover/branch.cpp, push_to_pair.cpp on the sweep of discover.cpp:84-197).  discover_tags_csv
 * null or "" = trace every assembly. */
int bg_discover_branch(bg_ctx* ctx, const bg_seqset* ss, const bg_readmap* rm, bg_asm_list* l, uint32_t min_overlap,
                       const char* tag, const char* discover_tags_csv);
int bg_discover_push_to_pair(bg_ctx* ctx, const bg_seqset* ss, const bg_readmap* rm, bg_asm_list* l,
                             uint32_t min_overlap, int64_t max_pair_distance, const char* tag,
                             const char* discover_tags_csv);

/* ---- row f.2, first slice: the production "push" tracer (modules/variants/tracer.cpp) --------------------------
 * Device part 1: tracer::advance_read_ahead_to (tracer.cpp:346-400) over one extent: d_entry[i] = the seqset entry the
 * read-ahead range is after base i when it is exactly one full entry, else ~0.  dup_first reproduces the double push
 * after tracer::skip_ahead_to (tracer.cpp:327-345). */
int bg_ref_walk_entries(bg_ctx* ctx, const bg_seqset* ss, const uint8_t* d_codes, uint64_t n, int dup_first,
                        uint64_t* d_entry);
/* Device part 2: the inner loop of tracer::expand_next_path (tracer.cpp:545-620) for a batch of paths, as far as it
 * depends on the seqset only. */
typedef struct bg_trace_query {
  uint64_t begin, end;   /* the path's range */
  uint64_t loop_check;   /* next_path::loop_check_seqset_id (~0 = none) */
  uint32_t size;
  uint32_t min_ctx;      /* assemble_options::min_overlap + next_path::pushed_since_read */
  uint32_t flags;        /* bit 0: pushed_any */
  uint32_t max_steps;
} bg_trace_query;
enum { BG_TRACE_AMBIGUOUS = 0, BG_TRACE_AT_READ = 1, BG_TRACE_STEPPED_ON_READ = 2, BG_TRACE_STEP_CAP = 3 };
typedef struct bg_trace_reply {
  uint64_t begin, end;   /* the range after the steps taken */
  uint64_t pb[4], pe[4]; /* where it stopped: the four pushes (valid_mask says which exist; loops removed) */
  uint32_t ps[4];
  uint32_t size, n_steps, reason, valid_mask;
} bg_trace_reply;
int bg_tracer_extend_batch(bg_ctx* ctx, const bg_seqset* ss, const bg_readmap* rm, uint64_t n,
                           const bg_trace_query* d_queries, uint32_t step_cap, bg_trace_reply* d_replies,
                           uint32_t* d_steps /* n * step_cap: base | size << 8 per step */);
/* Host part (csrc/tracer.cpp): reversable_tracer::assemble(start, limit) with the push tracer
 * (variants/reversable_tracer.cpp:76-145 -> tracer::assemble, tracer.cpp:195-318) followed by pair_counter
 * (pair_counter.cpp), with the anchor_dropper (anchor_drop.cpp) between them.  Fields = assemble_options
 * (assemble.h:475-720). */
typedef struct bg_trace_options {
  int64_t ambiguous_branch_cost, rejoin_local_cost, base_cost, increase_max_between_pair_cost, max_cost, pairs_used_cost,
      traverse_ref_cost, decrease_overlap_cost, size_change_cost, dead_end_cost, anchor_drop_score, pair_match_score,
      min_overlap_score;
  uint64_t max_pairs_per_read, max_next_paths, max_ambiguous_bases, min_overlap, read_ahead_distance, scaffold_split_size,
      max_branches_between_pairs, max_rejoins, max_bases_between_pairs, max_search_steps, max_ambiguous_search_steps,
      initial_search_steps, max_search_steps_per_read, max_pair_distance;
  int forward_pairs_face_inward, ignore_ambiguous_ref_pairs, trace_ambiguous_ref, trace_dead_ends, only_trace_forward,
      trace_reference_assemblies;
  double anchor_drop_size_multiplier;
  /* the anchor_dropper and aligner steps (csrc/align.cpp) */
  uint64_t min_anchor_drop_overlap;                 /* assemble.h:504 (15)                                            */
  int32_t ref_align_factor, max_ref_align_bases;    /* assemble.h:677-682 (4, 30)                                     */
  int32_t anchor_drop;            /* 1: run the anchor_dropper between tracer and pair_counter, as
                                     reversable_tracer.cpp:124-125 always does; 0 hands out the tracer's raw
                                     dead-end assemblies (right_anchor_len == 0) -- a test hook                       */
  int32_t max_anchor_kmer_size;   /* anchor_dropper::set_max_kmer_size_for_testing (anchor_drop.h:15-19); 31          */
  int32_t output_ml_features;     /* assemble.h:730: bg_assemblies_process records every discovered assembly's ML
                                     features behind the sorter (add_ml_features, biograph_variants.cpp:746-779)      */
  /* the pop tracer (bg_pop_trace) */
  uint64_t min_pop_overlap;         /* assemble.h:523 (15)   */
  uint64_t pop_tracer_offset_slop;  /* assemble.h:527 (1000) */
  int64_t min_pair_distance;        /* assemble.h:588 (100)  */
} bg_trace_options;
void bg_trace_options_default(bg_trace_options* o);
typedef struct bg_trace_stats {  /* assemble_stats of the tracer + the device rounds this implementation took */
  uint64_t traces, ref_reads, ambiguous_ref_reads, step_count, output_count, found_pairs, matched_pairs, max_branch_cost,
      suboptimal_path_prune, next_paths_too_big, extend_ambiguous_rejoin, search_not_fast_enough, too_many_ambiguous_bases,
      too_many_steps, too_many_ambiguous_steps, too_far_without_pair, exceeded_branch_limit, prune_ambiguous_ref,
      device_rounds, device_queries;
} bg_trace_stats;
typedef struct bg_trace_assembly {  /* variants::assembly as the tracer and pair_counter fill it (assemble.h:285-380) */
  int64_t left_offset, right_offset, score;
  int32_t left_anchor_len, right_anchor_len;
  const char* seq;
  uint64_t seq_len;
  uint32_t min_overlap, trace_steps, left_anchor_ambiguous_bases;
  int32_t matches_reference;
  const uint32_t *rc_read_ids, *left_pair_matches, *right_pair_matches;  /* rc_read_ids is empty after pair_count */
  uint64_t n_rc_read_ids, n_left_pair_matches, n_right_pair_matches;
  uint64_t assembly_id;  /* assemble.h:295: shared by the sections cut from one traced assembly; 0 in bg_trace_result_add
                          * = allocate a fresh one (assembly's constructor, assemble.cpp:1248) */
  int32_t from_pop;      /* assembly::tags: 1 = {"POP"} (pop_tracer.cpp:618), 0 = {"PUSH"} (tracer.cpp:1165) */
} bg_trace_assembly;
typedef struct bg_trace_result bg_trace_result;
/* Extents: the N-free stretches of the scaffold in scaffold coordinates, ascending; end_pos = scaffold::end_pos().
 * d_ref_map = bg_ref_map_build over the whole reference.  rev_comp: trace the reverse strand and report in forward
 * coordinates (reverse_assembly_in_place, assemble.cpp:933-960).  limit_offset - start_offset <= scaffold_split_size. */
int bg_push_trace(bg_ctx* ctx, const bg_seqset* ss, const bg_readmap* rm, const uint8_t* d_ref_map, uint64_t n_extents,
                  const int64_t* ext_offset, const char* const* ext_seq, const uint64_t* ext_len, int64_t end_pos,
                  int64_t start_offset, int64_t limit_offset, int rev_comp, int pair_count,
                  const bg_trace_options* options /* NULL = defaults */, bg_trace_result** out);
/* The same as a session: the host copies of the seqset / readmap / ref_map arrays the tracer consults are made once
 * (a trace_ref object and its assemble_options), bg_tracer_trace is one work item (scaffold piece x direction). */
typedef struct bg_tracer bg_tracer;
/* d_ref_map must stay valid until the session (and every clone of it) is closed: the read-ahead tables of a work item are
 * built on the device from it. */
int bg_tracer_open(bg_ctx* ctx, const bg_seqset* ss, const bg_readmap* rm, const uint8_t* d_ref_map,
                   const bg_trace_options* options /* NULL = defaults */, bg_tracer** out);
/* The extents a session is given must not change while it uses them: a session keeps its own checked copy of the scaffold
 * of its last work item, and a call that passes the same pointers, lengths and offsets again is taken to pass that
 * scaffold (a sample of the bytes is compared; anything else about the extents is checked in full). */
int bg_tracer_trace(bg_tracer* t, uint64_t n_extents, const int64_t* ext_offset, const char* const* ext_seq,
                    const uint64_t* ext_len, int64_t end_pos, int64_t start_offset, int64_t limit_offset, int rev_comp,
                    int pair_count, bg_trace_result** out);
/* The pop tracer over one work item (variants/pop_tracer.cpp:55-653 behind reversable_tracer.cpp:49-74,118-126; wired as
 * trace_ref::execute_work does, trace_ref.cpp:141-221; `biograph discovery --enable-pop-tracer`, on by default): the
 * anchor drops the push tracer reported while it produced `push[0 .. n_push)` (bg_tracer_trace results of THIS work item,
 * in the order they were traced: forward, then reverse -- each remembers its drops) and the mates of the reads along
 * [start_offset, limit_offset) become "fronts"; left-anchored drops lose their first base pass by pass
 * (seqset_range::pop_front on the device, all poppers of a pass in one launch) until what is left is a prefix of a
 * front.  Output as it reaches the scaffold's pipeline: sorter(left offset) -> pair_counter -> forward coordinates;
 * assemblies carry from_pop = 1.  rev_comp: the pop tracer that runs on the reverse strand (trace_ref's rc_pop). */
int bg_pop_trace(bg_tracer* t, uint64_t n_extents, const int64_t* ext_offset, const char* const* ext_seq,
                 const uint64_t* ext_len, int64_t end_pos, int64_t start_offset, int64_t limit_offset, int rev_comp,
                 const bg_trace_result* const* push, uint32_t n_push, bg_trace_result** out);
/* Number of anchor drops a bg_tracer_trace result remembers for bg_pop_trace. */
uint64_t bg_trace_result_anchor_drops(const bg_trace_result* r);
/* bg_tracer_trace with the host steps behind the tracer (anchor_dropper, pair_counter, the turn back to forward coordinates) left
 * to a helper thread of the session: the call returns when the tracer itself is done and the session can take its next work
 * item (trace_ref runs its work items on a thread pool, trace_ref.cpp:105-140; here a session overlaps two of them).  The result
 * must go through bg_trace_result_wait before any other use (bg_trace_result_free waits by itself; the anchor drops are complete
 * at once); the session must stay open until then.  Same result as bg_tracer_trace. */
int bg_tracer_trace_async(bg_tracer* t, uint64_t n_extents, const int64_t* ext_offset, const char* const* ext_seq,
                          const uint64_t* ext_len, int64_t end_pos, int64_t start_offset, int64_t limit_offset, int rev_comp,
                          int pair_count, bg_trace_result** out);
int bg_trace_result_wait(bg_trace_result* r);
/* A second session for another host thread: its own context (stream, allocator), the read-only host tables shared with
 * `from`, the same options.  Sessions may trace concurrently, one thread per session; `from` may be closed first. */
int bg_tracer_clone(const bg_tracer* from, bg_ctx* ctx, bg_tracer** out);
void bg_tracer_close(bg_tracer* t);
uint64_t bg_trace_result_size(const bg_trace_result* r);
int bg_trace_result_get(const bg_trace_result* r, uint64_t i, bg_trace_assembly* out);  /* pointers live as long as r */
int bg_trace_result_stats(const bg_trace_result* r, bg_trace_stats* out);
void bg_trace_result_free(bg_trace_result* r);

/* ---- row f.2, second slice (csrc/align.cpp): the head of the serial pipeline every tracer's output runs through
 * (variants/pipeline.cpp:40-57 and add_standard_variants_pipeline :92-95), as steps over bg_trace_result.
 * bg_assemblies_process takes the results of the work items of ONE scaffold (all pieces, both directions; forward
 * coordinates), runs the chosen steps in the pipeline's order and returns what leaves the last one:
 *   BG_STEP_ANCHOR_DROP  anchor_dropper (anchor_drop.cpp) -- bg_tracer_trace has run it already unless
 *                        options.anchor_drop was 0; for hand-made input (align_test.cpp:31-52)
 *   BG_STEP_SORT         sorter(canon_assembly_order) (sort.cpp:6-13, assemble.cpp:1029-1085); stable
 *   BG_STEP_TRIM_REF     ref_trimmer (trim_ref.cpp): drops reference assemblies, strips the anchors
 *   BG_STEP_DEDUP        deduper (dedup.cpp:7-144): merges assemblies with the same variant part
 *   BG_STEP_ALIGN        aligner (align.cpp:11-385): fills each assembly's aligned variants (bg_trace_result_var)
 *   BG_STEP_SPLIT        align_splitter (align.cpp:387-460): cuts into reference and variant sections
 *   BG_STEP_NORMALIZE    normalizer (normalize.cpp:7-126): shifts a variant left through repeats
 *   BG_STEP_EXACT_DEDUP  exact_deduper (dedup.cpp:146-216): one assembly per (left, right, seq, is-reference)
 *   BG_STEP_VCF_PAD      vcf_padder (normalize.cpp:128-162): inserts and deletions take the reference base to their left
 * Steps other than SORT need their input sorted by left offset, as the reference's do; a condition the reference
 * CHECK-aborts on (anchors that do not match the reference, unsorted input ...) fails the call with BG_EINVAL.
 * Extents as for bg_push_trace.  Host only: no bg_ctx, no device work. */
#define BG_STEP_ANCHOR_DROP 1u
#define BG_STEP_SORT 2u
#define BG_STEP_TRIM_REF 4u
#define BG_STEP_DEDUP 8u
#define BG_STEP_ALIGN 16u
#define BG_STEP_SPLIT 32u
#define BG_STEP_NORMALIZE 64u
#define BG_STEP_EXACT_DEDUP 128u
#define BG_STEP_VCF_PAD 256u
#define BG_STEP_ALL 511u
/* what pipeline.cpp:40-57,92-98 runs between the tracers and the coverage / genotype steps */
#define BG_STEP_PIPELINE (BG_STEP_SORT | BG_STEP_TRIM_REF | BG_STEP_DEDUP | BG_STEP_ALIGN | BG_STEP_SPLIT | \
                          BG_STEP_NORMALIZE | BG_STEP_EXACT_DEDUP | BG_STEP_VCF_PAD)
int bg_assemblies_process(const bg_trace_result* const* in, uint32_t n_in, uint64_t n_extents, const int64_t* ext_offset,
                          const char* const* ext_seq, const uint64_t* ext_len, int64_t end_pos,
                          const bg_trace_options* options /* NULL = defaults */, uint32_t steps, bg_trace_result** out);
/* The same steps fed piece by piece -- behind a scaffold's tracers instead of after them (the reference's pipeline of a
 * scaffold also takes its work items' output as they finish and sorts at the flush, pipeline.cpp:22-54, sort.cpp:6-13).
 * steps: as above, with BG_STEP_SORT and without BG_STEP_ANCHOR_DROP.  bg_pipeline_feed takes the results of the next
 * pieces, in the order bg_assemblies_process would be given them, and a mark: the caller's promise that no later input
 * has min(left_offset, right_offset) before it (a piece's assemblies cannot lie further before the piece than the tracers
 * read the reference: read_ahead_distance + max_pair_distance + two read lengths).  Everything held before the mark is
 * sorted and goes through the steps.  An input behind a mark already given breaks the promise: *behind = 1, the pipeline
 * is of no further use, and the caller runs bg_assemblies_process over all results instead.  With the promise kept,
 * bg_pipeline_finish hands out what bg_assemblies_process would: assembly for assembly.  The extents stay the caller's
 * and must outlive the pipeline.  Host only. */
typedef struct bg_pipeline bg_pipeline;
int bg_pipeline_open(uint64_t n_extents, const int64_t* ext_offset, const char* const* ext_seq, const uint64_t* ext_len,
                     int64_t end_pos, const bg_trace_options* options /* NULL = defaults */, uint32_t steps, bg_pipeline** out);
int bg_pipeline_feed(bg_pipeline* p, const bg_trace_result* const* in, uint32_t n_in, int64_t mark, int* behind);
/* what has left the steps since the last take (or the open), as a result of its own; bg_pipeline_finish hands out the rest */
int bg_pipeline_take(bg_pipeline* p, bg_trace_result** out);
uint64_t bg_pipeline_ready(const bg_pipeline* p);  /* how many assemblies a take would hand out now */
/* The reference's normalizer holds an extent's variants until the extent is through (a variant may move left through a repeat of
 * any length): a chromosome in one extent leaves the steps only at the finish.  With a margin (before the first feed) what lies
 * more than `margin` bases before the variant coming in leaves the normalizer at once; a later variant that moves to before
 * that breaks the promise like an input behind a mark: bg_pipeline_feed sets *behind, bg_pipeline_finish hands out NULL. */
int bg_pipeline_release_early(bg_pipeline* p, int64_t margin);
int bg_pipeline_finish(bg_pipeline* p, bg_trace_result** out);
void bg_pipeline_close(bg_pipeline* p);
/* Hand-made input for the steps (the reference's unit tests build assemblies by hand): an empty result, and a copy
 * of *a appended to it. */
int bg_trace_result_new(bg_trace_result** out);
int bg_trace_result_add(bg_trace_result* r, const bg_trace_assembly* a);
/* ---- row f.2, third slice (csrc/coverage.cpp): calc_coverage (with path_group), simple_genotype_filter and the VCF
 * record line (variants/pipeline.cpp:99-101, calc_coverage.cpp, path_group.cpp, simple_genotype_filter.cpp,
 * ploidless_vcf_export.cpp:12-166).  bg_assemblies_coverage takes the sections that leave BG_STEP_PIPELINE for one
 * scaffold (sorted by left offset), drops the reference sections and gives every variant section its interbase read
 * and pair coverage (seq_len + 1 values), the depth of the reference beside it and of the other alleles; with
 * genotype_filter != 0 the simple_genotype_filter runs too (strand count, genotype quality; failing assemblies are
 * dropped).  The path groups' walks run on the device (bg_walk_chains), one call per stretch of bases. */
typedef struct bg_coverage_options {
  uint32_t max_coverage_paths;             /* assemble.h:715 (4)    */
  int32_t penalize_directional_coverage;   /* assemble.h:719 (1)    */
  int32_t simple_genotype_filter;          /* assemble.h:607 (1)    */
  int32_t min_pair_depth, min_read_depth;  /* assemble.h:707-710 (0, 1) */
  uint32_t vcf_sv_size_threshold;          /* assemble.h:724 (20)   */
  double min_depth_portion;                /* assemble.h:704 (.23)  */
  double min_avg_pair_depth;               /* assemble.h:711 (1)    */
  int32_t output_ml_features;              /* assemble.h:730 (0): the LASCORE ... NUMASM columns of the VCF record */
  int32_t rvg_exclude;                     /* assemble.h:699 (0; biograph discovery --rvg-exclude defaults to true,
                                            * biograph_variants.cpp:194): run rvg_exclude behind the genotype filter */
} bg_coverage_options;
void bg_coverage_options_default(bg_coverage_options* o);
int bg_assemblies_coverage(bg_ctx* ctx, const bg_seqset* ss, const bg_readmap* rm, const bg_trace_result* in,
                           uint64_t n_extents, const int64_t* ext_offset, const char* const* ext_seq,
                           const uint64_t* ext_len, int64_t end_pos, const bg_coverage_options* options /* NULL = defaults */,
                           int genotype_filter, bg_trace_result** out);
/* The same as a session: the readmap columns calc_coverage reads are copied to the host once, bg_coverage_run is one
 * scaffold (a reference has thousands).  One run at a time per session. */
typedef struct bg_coverage bg_coverage;
int bg_coverage_open(bg_ctx* ctx, const bg_seqset* ss, const bg_readmap* rm, bg_coverage** out);
int bg_coverage_run(bg_coverage* c, const bg_trace_result* in, uint64_t n_extents, const int64_t* ext_offset,
                    const char* const* ext_seq, const uint64_t* ext_len, int64_t end_pos,
                    const bg_coverage_options* options /* NULL = defaults */, int genotype_filter, bg_trace_result** out);
/* A scaffold's sections in stretches, in order (bgbinary discovery: as bg_pipeline_take hands them out behind the tracers):
 * the islands a stretch completes -- runs of variant sections further than a read's length from the next (calc_coverage.cpp:
 * 345-360) -- are covered at once, the open one waits.  in may be NULL (nothing new).  last != 0: the scaffold is through;
 * the open island, then sort, genotype filter and rvg_exclude over everything, *out = what bg_coverage_run would give for the
 * concatenation.  One scaffold at a time per session; bg_coverage_discard forgets a scaffold that will not be finished. */
int bg_coverage_feed(bg_coverage* c, const bg_trace_result* in, uint64_t n_extents, const int64_t* ext_offset,
                     const char* const* ext_seq, const uint64_t* ext_len, int64_t end_pos, const bg_coverage_options* options,
                     int genotype_filter, int last, bg_trace_result** out);
void bg_coverage_discard(bg_coverage* c);
void bg_coverage_close(bg_coverage* c);
typedef struct bg_assembly_coverage {  /* assembly::coverage ... genotype_quality (assemble.h:330-370) */
  const int32_t *coverage, *pair_coverage;
  uint64_t n_coverage, n_pair_coverage;
  int32_t has_coverage, other_depth, other_pair_depth, ref_depth, strand_count;
  double genotype_quality;
} bg_assembly_coverage;
int bg_trace_result_coverage(const bg_trace_result* r, uint64_t i, bg_assembly_coverage* out);  /* pointers live as long as r */
int bg_trace_result_set_coverage(bg_trace_result* r, uint64_t i, const bg_assembly_coverage* in);  /* for hand-made input */
/* rvg_exclude (rvg_exclude.cpp:5-67; pipeline.cpp:102-104) as a step of its own, host only: a non-structural variant
 * section (both alleles below vcf_sv_size_threshold) with a position of zero pair coverage and no other_pair_depth is
 * dropped unless a structural section with the same assembly_id passes through.  `in` sorted by left offset. */
int bg_assemblies_rvg_exclude(const bg_trace_result* in, const bg_coverage_options* options /* NULL = defaults */,
                              bg_trace_result** out);
/* assembly_ml_features (assemble.h:191-199) + the number of assemblies merged into this one (deduper, exact_deduper) */
typedef struct bg_assembly_ml {
  int32_t has_ml, score, refspan, lanch, ranch;
  float refgc, altgc;
  uint64_t alt_len;
  uint32_t merged;
} bg_assembly_ml;
int bg_trace_result_ml(const bg_trace_result* r, uint64_t i, bg_assembly_ml* out);
int bg_trace_result_set_ml(bg_trace_result* r, uint64_t i, const bg_assembly_ml* in);
/* One VCF record line of ploidless_vcf_export::on_assembly for assembly i (nothing for a reference section: *len = 0);
 * assembly_id fills the PI column.  The line (with its newline) is *len bytes; at most cap - 1 of them are copied to buf. */
int bg_vcf_record(const bg_trace_result* r, uint64_t i, const char* scaffold_name, uint64_t n_extents,
                  const int64_t* ext_offset, const char* const* ext_seq, const uint64_t* ext_len, int64_t end_pos,
                  const bg_coverage_options* options, uint64_t assembly_id, char* buf, uint64_t cap, uint64_t* len);
/* path_group::add_base for a whole sequence: the ranges of n_chains paths after each of len bases (2-bit codes), every
 * step a push_front_drop(complement, 0).  h_start / h_out: (begin, end, size) as three u64 per range, HOST buffers (the
 * caller is a host state machine); h_out[(t * n_chains + i) * 3 ..] = chain i after base t. */
int bg_walk_chains(bg_ctx* ctx, const bg_seqset* ss, uint32_t n_chains, const uint64_t* h_start, const uint8_t* h_codes,
                   uint32_t len, uint64_t* h_out);
/* The same for chains of many groups at once (calc_coverage advances independent stretches of the scaffold in lock
 * step): every chain names its own bases and where its ranges go; one upload, one launch, one download. */
typedef struct bg_walk_chain {
  uint64_t begin, end;      /* start range                                                     */
  uint64_t code_offset;     /* its bases: h_codes[code_offset .. code_offset + len)            */
  uint64_t out_offset;      /* range after base t: h_out[out_offset + t * out_stride .. + 3)   */
  uint64_t out_stride;      /* in u64 words, >= 3; h_out is rewritten as a whole: words no chain names come back 0 */
  uint32_t size, len;
} bg_walk_chain;
int bg_walk_chains_batch(bg_ctx* ctx, const bg_seqset* ss, uint32_t n_chains, const bg_walk_chain* h_chains,
                         const uint8_t* h_codes, uint64_t n_codes, uint64_t* h_out, uint64_t out_words);
/* Chains of seqset_range::push_front (seqset.cpp:319-338) from the empty context: chain i pushes
 * h_codes[h_offsets[i] .. h_offsets[i+1]) in order and stops at the first base after which no entry is left;
 * h_out[3 i ..] = (begin, end, size) of the last valid range (pop_tracer.cpp:258-266).  HOST buffers. */
int bg_push_front_chains(bg_ctx* ctx, const bg_seqset* ss, uint32_t n, const uint64_t* h_offsets, const uint8_t* h_codes,
                         uint64_t* h_out);
/* What the anchor_dropper of a bg_tracer_trace / bg_push_trace call could not anchor on the right (variants::half_aligned_assembly,
 * assemble.h:449-465; assemble_options::report_half_aligned_func, anchor_drop.cpp:301-329, turned into forward coordinates for a
 * reverse trace as reversable_tracer.cpp:20-27 does): right_anchor 0 = the sequence starts at `offset` and runs right, 1 = it ends
 * there.  seq points into r.  `biograph discovery --half-aligned-out` (biograph_variants.cpp:175,695-710) writes these. */
uint64_t bg_trace_result_half_aligned_count(const bg_trace_result* r);
int bg_trace_result_half_aligned(const bg_trace_result* r, uint64_t i, int64_t* offset, int* right_anchor, const char** seq,
                                 uint64_t* seq_len, uint64_t* assembly_id);
/* aligned_variants of assembly i (variants::aligned_var, assemble.h:124-152): filled by BG_STEP_ALIGN, consumed by
 * BG_STEP_SPLIT.  seq points into r. */
uint64_t bg_trace_result_var_count(const bg_trace_result* r, uint64_t i);
int bg_trace_result_var(const bg_trace_result* r, uint64_t i, uint64_t j, int64_t* left_offset, int64_t* right_offset,
                        const char** seq, uint64_t* seq_len);


/* ---- Merge: `biograph merge` = seqset_flat + make_mergemap + seqset_merger
 * (modules/biograph/biograph_merge.cpp:189-290, bio_base/make_mergemap.cpp:150-220,
 * seqset_merger.cpp:50-185).  The merged seqset holds the prefix-free maxima of the union of the
 * inputs' entries.  Inputs built in this process use their resident keys; inputs opened from files are
 * flattened by FM navigation first (seqset_flat).  d_maps (may be null) = one device array per input of
 * bg_seqset_size(in[i]) u64: the merged id of every input entry, i.e. seqset_mergemap's
 * find_count(entry) (the mergemap bit vector is the set of those ids).  NB the reference's sweep may put
 * the prev bit of an entry whose tail is a proper prefix of several merged entries on a later one of
 * them, depending on its parallel_for chunking (seqset_merger.cpp:89-106); here it is always the first
 * (the builder's rule, builder.cpp:85-108). */
int bg_seqset_merge(bg_ctx* ctx, uint32_t n_in, const bg_seqset* const* in, bg_seqset** out,
                    uint64_t* const* d_maps);
/* make_readmap::create_from_fast_migrate (bio_mapred/make_readmap.cpp:455-519): the same read ids
 * (lengths, mate loop pointers, orientation) on the merged seqset, entries translated through d_map. */
int bg_readmap_migrate(bg_ctx* ctx, const bg_readmap* old_rm, const uint64_t* d_map, const bg_seqset* merged,
                       bg_readmap** out);
/* bg_create() plus the sample's readmap: the whole data path of SEQSETMain::run
 * (biograph_create.cpp:431-831: kmerize, correct, make_seqset, make_readmap). */
int bg_create_sample(bg_ctx* ctx, const uint8_t* d_seq, const uint64_t* d_offs, uint64_t n_reads,
                     const bg_create_params* p, int is_paired, bg_seqset** out, bg_readmap** out_rm,
                     bg_build_stats* stats);

/* ---- Sharded build (one process per GPU).  The library never communicates: these
 * entry points expose the phases of bg_create() at the exchange points; the host moves the buffers
 * with RCCL (torch.distributed all_to_all / all_gather) — see biograph_amd/sharded.py.
 * "Rows" are fixed-width records of u64 words on the device. */
#define BG_PART_KMER_HASH 0  /* 1-word rows = canonical k-mers; dest = hash(kmer) mod nparts (the 256-way
                                kmer_partition of kmer_counter.h:167-178); all-ones rows are dropped   */
#define BG_PART_PREFIX 1     /* rows of W+1 words; h_params = {prefix_bases, nparts-1 ascending bin
                                splitters}; dest = #splitters <= leading-bases bin
                                (part_repo::partition_for_sequence, part_repo.cpp:32-45)              */
#define BG_PART_KEY_BOUNDS 2 /* rows of W+1 words (last = len | base<<8); h_params = nparts rows: last
                                entry of each shard (last word ~0 = empty); dest = first shard whose
                                last entry >= row                                                       */
#define BG_PART_ROW_DEST 3   /* the row names its destination itself: bits 10..16 of its last word (127 = nowhere) */
/* ---- Communicator and the N-rank create (csrc/comm.hip, csrc/sharded.hip): `bgbinary create --gpus N`.
 * The reference's SEQSETMain::run (modules/biograph/biograph_create.cpp:431-811) runs its partitions
 * (part_repo::partition_for_sequence, modules/build_seqset/part_repo.cpp:32-45) as `parallel_for` workers over shared disk
 * files; here one process per GPU owns its partitions in HBM and what the reference moves through files moves through
 * RCCL over xGMI.  One bg_comm per process: rank r of `world` on the context's device.
 *   backend "rccl": bg_comm_unique_id() on one rank (ncclGetUniqueId), its 128 bytes handed to the others by the launcher
 *                   (pipe, file, store), bg_comm_init(..., id, "rccl") on every rank (ncclCommInitRank).  Pieces of
 *                   an exchange are at most 1 GiB - 4 KiB (BG_COMM_MAX_PIECE), larger exchanges take several groups.
 *   backend "shm":  host-staged through POSIX shared memory named by `id` (a short string): several ranks on ONE GPU
 *                   (tests, boxes without xGMI) -- RCCL refuses two ranks on a device.
 * A rank alone turns every collective into a copy unless BG_COMM_NO_SHORTCUT=1 (then its pieces go through the backend
 * to itself).  Collectives are queued on the context's stream; host-valued ones return when they have landed. */
typedef struct bg_comm bg_comm;
#define BG_COMM_ID_BYTES 128
typedef struct bg_comm_stats {
  uint64_t collectives;            /* exchanges, gathers and reductions issued                              */
  uint64_t bytes_through_backend;  /* bytes this rank sent through RCCL (or staged through shared memory)   */
  uint64_t bytes_kept_local;       /* pieces a rank keeps: device copies                                     */
  double seconds_in_calls;         /* host time inside the communicator's calls                              */
  int backend;                     /* 0 rccl, 1 shm, 2 none (a rank alone, short-cut)                        */
  int rccl_version;                /* ncclGetVersion (0 unless backend 0)                                    */
} bg_comm_stats;
int bg_comm_unique_id(char* h_id_out /* BG_COMM_ID_BYTES */);
int bg_comm_init(bg_ctx* ctx, int rank, int world, const char* id, const char* backend /* "rccl" | "shm" */, bg_comm** out);
void bg_comm_free(bg_comm* c);
int bg_comm_rank(const bg_comm* c);
int bg_comm_world(const bg_comm* c);
int bg_comm_get_stats(const bg_comm* c, bg_comm_stats* out);
/* rows of row_bytes travel as h_send_rows[d] rows to rank d (pieces back to back in d_send, rank order) and arrive as
 * h_recv_rows[s] rows from rank s (back to back in d_recv).  _all_to_all_v learns the receive counts itself (one small
 * all-gather); _exchange takes them from the caller (the way back of an earlier exchange). */
int bg_comm_all_to_all_v(bg_comm* c, const void* d_send, const uint64_t* h_send_rows, uint64_t row_bytes, void* d_recv,
                         uint64_t recv_cap_rows, uint64_t* h_recv_rows);
int bg_comm_exchange(bg_comm* c, const void* d_send, const uint64_t* h_send_rows, const uint64_t* h_recv_rows, uint64_t row_bytes,
                     void* d_recv);
int bg_comm_all_gather_host(bg_comm* c, const void* h_mine, uint64_t bytes, void* h_all /* world x bytes */);
int bg_comm_all_gather(bg_comm* c, const void* d_send, const uint64_t* h_bytes /* world: bytes of every rank */, void* d_recv);
int bg_comm_all_reduce_sum_u64(bg_comm* c, uint64_t* d_buf, uint64_t n);
int bg_comm_barrier(bg_comm* c);

/* bg_create_sample() for ONE sample whose reads are spread over the ranks of `comm` (each rank passes its share: whole
 * pairs, any split; read ids follow rank order): SEQSETMain::run (biograph_create.cpp:431-831) with every structure owned by
 * one rank -- k-mers by the hash partition of their minimiser (kmer_counter.h:167-178), cover index, records and entries
 * by their leading bases (part_repo.cpp:32-45), readmap rows by their entry.  Collective: every rank calls it.  The
 * result is the rank's 512-aligned stretch of the seqset (bg_sharded_slice) and its stretch of the readmap's columns;
 * bg_sharded_write lets all ranks write ONE seqset container in place (spiral_file_mmap.cpp:350-457) and gathers the
 * readmap on rank 0, which writes it.  The containers are byte-identical to the single-GPU build's (except file_info.json). */
typedef struct bg_sharded bg_sharded;
typedef struct bg_sharded_stats {
  bg_build_stats build;                    /* of the whole sample (n_tied_records: not collected)                  */
  uint64_t shard_entries;                  /* entries this rank owned before the slices were aligned               */
  uint64_t first_entry, slice_entries;     /* this rank's slice: [first_entry, first_entry + slice_entries)        */
  uint64_t n_read_ids, first_read_id, slice_read_ids;
  uint64_t correct_probes, tails_sent_in_full;
  uint32_t correct_steps;
} bg_sharded_stats;
int bg_create_sharded(bg_ctx* ctx, bg_comm* comm, const uint8_t* d_seq, const uint64_t* d_offs, uint64_t n_reads,
                      const bg_create_params* p, int readmap /* 0 none, 1 unpaired, 2 paired */, bg_sharded** out,
                      bg_sharded_stats* stats);
const bg_seqset* bg_sharded_slice(const bg_sharded* s);
int bg_sharded_write(bg_ctx* ctx, bg_comm* comm, const bg_sharded* s, const char* seqset_path, const char* readmap_path /* may be NULL */,
                     int argc, const char* const* argv, char* h_uuid37 /* may be NULL */, char* h_readmap_sha41 /* may be NULL */);
void bg_sharded_free(bg_sharded* s);

/* `biograph merge` over the ranks of `comm` (biograph_merge.cpp:189-290; seqset_flat / make_mergemap / seqset_merger as their
 * net semantics: the merged entries are the prefix-free maxima of the union): every rank holds ALL inputs (opened from their
 * files or built) and takes the stripe [n_i r / P, n_i (r + 1) / P) of input i's entries through the records exchange with
 * common splitters; the result is the rank's 512-aligned stretch of the merged seqset, for bg_sharded_write.  d_maps (may be
 * NULL, as may any of its n_in pointers): for every entry of this rank's stripe of input i, the global id of the merged
 * entry that holds it (make_mergemap.cpp:150-220) -- what bg_readmap_migrate_to takes once the stripes are put together.
 * Collective: every rank calls it with the same inputs. */
/* A shard whose rows came from a create (bg_records_set_meta before bg_entries_from_records) sets prev bits where its sorted
 * records meet (builder.cpp:85-108 without a search) and keeps which entries still have to send their
 * tail; the look-ups that decide a few more travel as rows (bg_seqset_cond_rows -> the shard their key falls into ->
 * bg_seqset_mark_need).  Then bg_seqset_tail_rows2_needed replaces bg_seqset_tail_rows2 (d_rows NULL: the count alone). */
void bg_records_set_meta(bg_records* r, int has_meta);
int bg_seqset_local_prev(const bg_seqset* ss);
int bg_seqset_cond_rows(const bg_seqset* ss, const uint64_t** d_rows, uint64_t* n_rows);
int bg_seqset_mark_need(bg_ctx* ctx, bg_seqset* ss, const uint64_t* d_rows, uint64_t n_rows);
int bg_seqset_tail_rows2_needed(bg_ctx* ctx, const bg_seqset* ss, uint32_t nparts, const uint64_t* h_bounds, uint64_t* d_rows,
                                uint64_t cap, uint64_t* n_out);

int bg_merge_sharded(bg_ctx* ctx, bg_comm* comm, uint32_t n_in, const bg_seqset* const* in, bg_sharded** out,
                     bg_sharded_stats* stats, uint64_t* const* d_maps);
/* bg_readmap_migrate against a merged seqset known by its size and uuid only (the sharded merge's output lives in slices). */
int bg_readmap_migrate_to(bg_ctx* ctx, const bg_readmap* old_rm, const uint64_t* d_map, uint64_t merged_entries,
                          const char* merged_uuid, bg_readmap** out);

/* ---- reference: what the discovery walks along ------------------------------------
 * bg_reference_open reads a BioGraph reference directory (karyotype.json + reference.ref, or reference.fasta
 * when the .ref is absent; modules/io/defaults.h:17-50, bio_base/reference.cpp:27-40), a bare .ref file
 * (bio_base/flat_ref.cpp:140-165: "SPR000", 2-bit flat genome, msgpack index) or a FASTA (flat_ref.cpp:18-24).
 * Scaffolds = chromosomes (reference_assembly::scaffold_order); supercontigs = their N-free stretches in flat
 * order (reference_assembly::supercontigs): scaffold index, offset inside it, length, flat position.
 * bg_reference_bases copies ASCII bases of the flat genome.  Host only: no bg_ctx needed. */
typedef struct bg_reference bg_reference;
int bg_reference_open(const char* path, bg_reference** out);
void bg_reference_free(bg_reference* ref);
uint64_t bg_reference_size(const bg_reference* ref);
uint32_t bg_reference_scaffold_count(const bg_reference* ref);
const char* bg_reference_scaffold_name(const bg_reference* ref, uint32_t i);
uint64_t bg_reference_scaffold_len(const bg_reference* ref, uint32_t i);
uint32_t bg_reference_supercontig_count(const bg_reference* ref);
int bg_reference_supercontig(const bg_reference* ref, uint32_t i, uint32_t* scaffold, uint64_t* offset, uint64_t* len,
                             uint64_t* flat);
int bg_reference_bases(const bg_reference* ref, uint64_t flat_pos, uint64_t n, char* out);

/* ---- reference.bwt on the host: bwt_file / bwt_range (modules/bio_base/bwt_file.h:9-66, bwt_file.cpp:20-112), what
 * python/biograph/reference.cpp:116-185 (Reference.find -> ReferenceContext) reads.  bg_bwt_open takes the file or the
 * reference directory that holds it.  bg_bwt_find = bwt_range::find over the whole table (an empty range when the
 * sequence does not occur); bg_bwt_match = bwt_range::get_match for one sorted suffix: its flat reference position. */
typedef struct bg_bwt bg_bwt;
int bg_bwt_open(const char* path, bg_bwt** out);
void bg_bwt_free(bg_bwt* bwt);
uint64_t bg_bwt_entries(const bg_bwt* bwt);
int bg_bwt_find(const bg_bwt* bwt, const char* seq, uint64_t len, uint64_t* begin, uint64_t* end);
int bg_bwt_match(const bg_bwt* bwt, uint64_t entry, uint64_t* flat_pos);

/* ---- `biograph reference` (modules/main/make_ref.cpp -> pipeline/build_reference_task.cpp): building a reference
 * directory.  bg_reference_build = IMPORT_FASTA + MAKE_FLAT on the host (bio_format/fasta_ref_importer.cpp,
 * bio_base/flat_ref.cpp:224-360): reads plain FASTA text and writes <refdir>/reference.fasta, karyotype.json and
 * reference.ref.  bg_bwt_build = MAKE_BWT on the GPU (bio_mapred/make_bwt.cpp:33-141): the suffix array of all extents
 * by prefix doubling and the .bwt file of bio_base/bwt_file.h (cent_mod = 64 in the reference). */
int bg_reference_build(const char* source_fasta, const char* refdir, uint64_t min_n_run /* 50 */);
typedef struct bg_bwt_stats { uint64_t n_entries, n_century; uint32_t doubling_rounds; } bg_bwt_stats;
int bg_bwt_build(bg_ctx* ctx, const uint8_t* d_codes /* flat reference, one code 0..3 per byte */, uint32_t n_extents,
                 const uint64_t* h_ext_flat, const uint64_t* h_ext_size, uint64_t cent_mod, const char* path,
                 bg_bwt_stats* stats /* may be NULL */);

/* Stable partition of n rows into nparts destination groups written back to back into d_out_rows;
 * h_counts[p] = rows of group p. */
int bg_partition_rows(bg_ctx* ctx, const uint64_t* d_rows, uint64_t n, int row_words, int mode,
                      uint32_t nparts, const uint64_t* h_params, uint64_t* d_out_rows, uint64_t* h_counts);

/* The rows whose flag byte is non-zero, in order, into d_out (room for cap rows; *n_out = how many). */
int bg_rows_select(bg_ctx* ctx, const uint64_t* d_rows, uint64_t n, int row_words, const uint8_t* d_flags, uint64_t* d_out,
                   uint64_t cap, uint64_t* n_out);

/* k-mer phases of bg_kmer_count(): windows -> keys; keys (after the exchange) -> kept k-mers of this
 * shard's hash range; all kept k-mers (after the all-gather) -> hash set. */
int bg_kmer_instances(bg_ctx* ctx, const bg_reads* reads, int k, uint64_t* n_out);
int bg_kmer_extract(bg_ctx* ctx, const bg_reads* reads, int k, uint64_t* d_keys, uint64_t cap);
/* bg_kmer_extract with the exchange partition fused in: the instances leave grouped by owner rank
 * (owner of a k-mer = top byte of kmer * 0x9E3779B97F4A7C15, times nparts, over 256 — whole buckets of
 * the counting partition, i.e. the role of the 256-way kmer_partition of kmer_counter.h:167-178);
 * h_counts[nparts] = rows per owner, consecutive in d_keys.  The all-ones rows of windows holding an N
 * travel with their bucket and are ignored by bg_kmer_count_keys. */
int bg_kmer_extract_by_owner(bg_ctx* ctx, const bg_reads* reads, int k, uint32_t nparts, uint64_t* d_keys,
                             uint64_t cap, uint64_t* h_counts);
int bg_kmer_count_keys(bg_ctx* ctx, uint64_t* d_keys /* clobbered */, uint64_t n, int k, uint32_t min_count,
                       bg_kmerset** out /* kmers+counts only, no hash */);
/* The same exchange with super-k-mers instead of k-mer instances (k >= 14; csrc/kmer_skm.hip): a record is 16
 * bytes = the bases of a run of windows of one read that share their minimiser, i.e. ~1.6 bytes per window
 * instead of 8 across the links.  Owner of a record = top byte of its bucket hash, times nparts, over 256.
 *   bg_skm_extract_by_owner: records of `reads` grouped by owner into d_recs (2 words each) with their origins
 *     (read << 8 | first window) in d_org, both with room for `cap` records (bg_skm_capacity);
 *     h_counts[nparts] = records per owner, *n_out = their sum.  n_windows = bg_kmer_instances().
 *   bg_skm_count: the records a rank owns (from all ranks, any order) -> its kept k-mers (kmers + counts, no
 *     hash) and, in d_bits[n] (may be NULL), for every record the windows whose k-mer is kept (bit t = window
 *     t; 0x80000000 = not known here, the record's bucket overflowed the counting table).
 *   bg_kmerset_attach_presence: the bits, sent back to the rank the records came from, become the per-read
 *     window presence of a hash-carrying k-mer set (unknown records are looked up in it), which lets
 *     bg_correct_reads(reads, ks) skip the look-up of every window (as after bg_kmer_count). */
int bg_skm_supported(int k);
int bg_skm_capacity(bg_ctx* ctx, const bg_reads* reads, int k, uint64_t n_windows, uint64_t* cap_out);
int bg_skm_extract_by_owner(bg_ctx* ctx, const bg_reads* reads, int k, uint64_t n_windows, uint32_t nparts,
                            uint64_t* d_recs, uint64_t* d_org, uint64_t cap, uint64_t* h_counts, uint64_t* n_out);
int bg_skm_count(bg_ctx* ctx, const uint64_t* d_recs, uint64_t n, int k, uint32_t min_count, bg_kmerset** out,
                 uint32_t* d_bits);
int bg_kmerset_attach_presence(bg_ctx* ctx, bg_kmerset* ks, const bg_reads* reads, const uint64_t* d_recs,
                               const uint64_t* d_org, const uint32_t* d_bits, uint64_t n);
int bg_kmerset_export(bg_ctx* ctx, const bg_kmerset* ks, uint64_t* d_kmers_out);
int bg_kmerset_from_kmers(bg_ctx* ctx, const uint64_t* d_kmers, uint64_t n, int k,
                          const bg_reads* reads_for_flags /* may be NULL */, bg_kmerset** out);

/* corrected reads travel as (words[n*W], len[n]) */
int bg_reads_words(const bg_reads* r);
int bg_reads_export(bg_ctx* ctx, const bg_reads* r, int W_out, uint64_t* d_words, uint16_t* d_len);
int bg_reads_import(bg_ctx* ctx, uint64_t n, int W, uint32_t max_len, const uint64_t* d_words,
                    const uint16_t* d_len, bg_reads** out);

/* suffix records of reads [read_lo, read_hi) of `reads`; the cover index spans all of `reads` */
int bg_suffix_records(bg_ctx* ctx, const bg_reads* reads, uint64_t read_lo, uint64_t read_hi, int cover_k,
                      bg_records** out);
uint64_t bg_records_count(const bg_records* r);
int bg_records_words(const bg_records* r);
int bg_records_rows(bg_ctx* ctx, const bg_records* r, uint64_t* d_rows /* n*(W+1): words, len */);
/* d_hist[4^bases] (u64, device): how many of the records 0, every, 2*every, ... start with each `bases`-base
 * prefix (every = 1: all of them) — what the ranks all-reduce to choose balanced splitters for the records
 * exchange (part_counts.h:8-45); a sample is enough to balance and costs 1/every of the global atomics. */
int bg_records_prefix_hist(bg_ctx* ctx, const bg_records* r, int bases, uint64_t every, uint64_t* d_hist);
int bg_records_from_rows(bg_ctx* ctx, const uint64_t* d_rows, uint64_t n, int W, bg_records** out);
/* the same from several row arrays (n_parts device pointers and row counts, host arrays): no concatenated copy */
int bg_records_from_row_parts(bg_ctx* ctx, uint32_t n_parts, const uint64_t* const* d_rows, const uint64_t* h_counts, int W,
                              bg_records** out);
void bg_records_free(bg_records* r);
/* sort + prefix-dedup of one shard's records -> entries (keys/sizes/shared, no prev bits); consumes r */
int bg_entries_from_records(bg_ctx* ctx, bg_records* r, bg_seqset** out, bg_build_stats* stats);

/* shard boundaries (part_repo.cpp:294-303, builder.cpp:225): rows of W+1 host words */
int bg_seqset_words(const bg_seqset* ss);
int bg_seqset_boundary_rows(bg_ctx* ctx, const bg_seqset* ss, uint64_t* h_first_row, uint64_t* h_last_row);
int bg_seqset_fix_boundary(bg_ctx* ctx, bg_seqset* ss, const uint64_t* h_prev_last_row,
                           const uint64_t* h_next_first_row);
/* prev bits across shards: tails s[1:] of own entries out, tails owned by this shard in */
/* Sharded merge (replaces make_mergemap, bio_base/make_mergemap.cpp:150-220, for inputs that live as shards):
 * d_rows = the rows this shard received, last word = length | input << 8; d_marks[input * n + e] = 1 when entry e of
 * the shard is the image of an entry of that input; h_carry[input] = 1 when an entry of the input maps to the first
 * entry of the next non-empty shard. */
int bg_seqset_mark_inputs(bg_ctx* ctx, const bg_seqset* shard, const uint64_t* d_rows, uint64_t n_rows,
                          uint32_t n_inputs, uint8_t* d_marks /* n_inputs * n */, uint32_t* h_carry /* n_inputs */);
/* The entries as rows of W+1 words (key words, length): what a sharded merge exchanges (replaces the per-entry
 * merge of biograph_merge.cpp:189-290 for seqsets that live as slices on several GPUs). */
int bg_seqset_entry_rows(bg_ctx* ctx, const bg_seqset* ss, uint64_t* d_rows /* n*(W+1) */);
int bg_seqset_tail_rows(bg_ctx* ctx, const bg_seqset* ss, uint64_t* d_rows /* n*(W+1) */);
int bg_seqset_apply_tails(bg_ctx* ctx, bg_seqset* ss, const uint64_t* d_rows, uint64_t n_rows);
/* The same with short tails: 3-word rows (the tail's two leading words, meta = len-1 | base << 8 | destination
 * shard << 10 | entry << 17; route them with BG_PART_ROW_DEST).  The receiving shard settles what its entries'
 * two-word prefixes decide and asks for the whole tail of the rest (d_need_full, sent back in order);
 * bg_seqset_tail_rows_full builds those (same meta) for bg_seqset_apply_tails.  40 % of the bytes of full tails. */
int bg_seqset_tail_rows2(bg_ctx* ctx, const bg_seqset* ss, uint32_t nparts, const uint64_t* h_bounds /* nparts x (W+1) */,
                         uint64_t* d_rows /* n*3 */);
int bg_seqset_apply_tails2(bg_ctx* ctx, bg_seqset* ss, const uint64_t* d_rows, uint64_t n_rows, uint8_t* d_need_full);
int bg_seqset_tail_rows_full(bg_ctx* ctx, const bg_seqset* ss, const uint64_t* d_short_rows, const uint8_t* d_need,
                             uint64_t n, uint64_t* d_rows_out, uint64_t cap, uint64_t* n_out);
/* final merge = concatenation of the shards' slices in rank order + seqset::finalize */
int bg_seqset_slice_export(bg_ctx* ctx, const bg_seqset* ss, uint8_t* d_sizes, uint8_t* d_shared,
                           uint64_t* d_bits /* 4 x ceil(n/64) */);
int bg_seqset_assemble(bg_ctx* ctx, uint32_t nparts, const uint64_t* h_part_n, const uint8_t* d_sizes,
                       const uint8_t* d_shared, const uint64_t* d_bits, uint32_t max_len, bg_seqset** out);

/* ---- A sharded seqset leaves as slices (no gather on one rank): after the prev bits are applied every rank
 * re-aligns its entries to multiples of 512 with its neighbours (sizes, shared and one mask byte per entry:
 * bit b = prev_b), builds its slice and writes its bytes of every container member in place.
 *   bg_seqset_slice_masks          per-entry arrays of a shard (device): sizes, shared, prev mask
 *   bg_seqset_slice_build          seqset::finalize (seqset.cpp:113-129, bitcount.cpp:84-124) of an aligned stretch;
 *                                  max_len = of the whole seqset (it fixes the packed widths)
 *   bg_seqset_slice_popcounts      set prev bits per base inside the slice
 *   bg_seqset_sharded_create_file  rank 0: container with room for every payload member (spiral_file_mmap.cpp:350-457);
 *                                  h_offsets[15] = file offsets of the members' data; writes `fixed`
 *   bg_seqset_slice_write          every rank: its bytes in place (first_entry % 512 == 0; inner slices hold a
 *                                  multiple of 512 entries; h_base_pop4 = set bits per base before first_entry) */
int bg_seqset_slice_masks(bg_ctx* ctx, const bg_seqset* ss, uint8_t* d_sizes, uint8_t* d_shared, uint8_t* d_mask);
int bg_seqset_slice_build(bg_ctx* ctx, uint64_t m, const uint8_t* d_sizes, const uint8_t* d_shared, const uint8_t* d_mask,
                          uint32_t max_len, bg_seqset** out);
int bg_seqset_slice_popcounts(const bg_seqset* ss, uint64_t* h_out4);
int bg_seqset_sharded_create_file(const char* path, uint64_t n_total, uint32_t max_len, const uint64_t* h_fixed5, int argc,
                                  const char* const* argv, uint64_t* h_offsets, char* h_uuid37 /* may be NULL: the uuid given to the seqset */);
int bg_seqset_slice_write(bg_ctx* ctx, const bg_seqset* slice, const char* path, const uint64_t* h_offsets,
                          uint64_t first_entry, int is_last, const uint64_t* h_base_pop4);

/* ---- Routed read correction of the sharded build: the kept k-mers stay with the rank that counted them (a
 * k-mer lives in ONE partition, kmer_counter.h:167-178) instead of being gathered on every rank.
 *   bg_kmerset_build_hash   owner: hash table over the kept k-mers bg_skm_count left
 *   bg_skm_lookup           owner: d_bits[i] bit t = window t of record i is kept (mode bit 0: just the records
 *                           bg_skm_count marked 0x80000000 — its counting table overflowed; mode bit 1: a record's
 *                           windows are only looked up as far as the first absent one — probe records, whose
 *                           windows the search consumes in order)
 *   bg_corr_begin           the rank that holds the reads: window presence of its reads from the bits its
 *                           records got back (all resolved), screen (correct_reads.cpp:154-178)
 *   bg_corr_step            fast_read_correct (fast_read_correct.cpp:16-182) of every pending read over the
 *                           probes it has; *probes = strings whose windows are the k-mers the searches missed
 *                           (a reads object: ship its super-k-mers like the reads', bg_skm_extract_by_owner)
 *   bg_corr_answers         the owners' bits for those records, in sent order
 *   bg_corr_finish          corrected reads + stats once no read is pending (same result as bg_correct_reads) */
typedef struct bg_correction bg_correction;
int bg_kmerset_build_hash(bg_ctx* ctx, bg_kmerset* ks);
int bg_skm_lookup(bg_ctx* ctx, const bg_kmerset* ks, const uint64_t* d_recs, uint64_t n, uint32_t* d_bits,
                  int mode);
int bg_corr_begin(bg_ctx* ctx, const bg_reads* reads, int k, const bg_correct_params* params, const uint64_t* d_recs,
                  const uint64_t* d_org, const uint32_t* d_bits, uint64_t n_recs, bg_correction** out);
int bg_corr_step(bg_ctx* ctx, bg_correction* c, uint64_t* n_pending_out, bg_reads** probes_out);
int bg_corr_answers(bg_ctx* ctx, bg_correction* c, const uint64_t* d_org, const uint32_t* d_bits, uint64_t n);
int bg_corr_finish(bg_ctx* ctx, bg_correction* c, bg_reads** out, bg_correct_stats* stats);
uint64_t bg_corr_probes(const bg_correction* c);
void bg_corr_free(bg_correction* c);

/* ---- Sharded exact-cover emission (csrc/shard_cover.hip): the cover index is range-partitioned by leading
 * bases like the suffix records (part_repo::partition_for_sequence, part_repo.cpp:32-45) instead of being
 * replicated.  Rows are W+1 words: W key words, last word = length (bits 0-7) | plain flag (bit 16) | tag
 * (bits 24-63); a row of length 0 is dropped by bg_partition_rows(BG_PART_PREFIX).
 *   bg_oriented_rows        2 rows per read of `reads` (forward, reverse complement; length 0 = dropped read)
 *   bg_rows_prefix_hist     sampled histogram of the rows' leading bases (splitters, part_counts.h:8-45)
 *   bg_cover_build          owner: index over the oriented rows it received (they are also its j = 0 records)
 *   bg_cover_test           owner: d_covered[i] = row i is a prefix of an indexed read (plain rows: never)
 *   bg_cover_query_*        sender: candidate suffixes [j_lo, j_hi) of the reads still uncovered, tagged;
 *                           answers (in sent order) -> which candidates the owner keeps as records (j below the
 *                           read's first covered suffix); plain = the untested tails (< K bases) of the reads
 *                           that stay uncovered after the last round (the last round must reach j_hi = 256). */
typedef struct bg_cover bg_cover;
typedef struct bg_cover_query bg_cover_query;
/* the oriented reads whose suffix 1 is covered (their whole-read record is their last), as rows of W + 1 words: the look-up
 * rows of bg_seqset_mark_need for a shard that sets prev bits where its records meet (d_rows NULL: the count alone) */
int bg_cover_query_first_covered_rows(bg_ctx* ctx, const bg_cover_query* q, uint64_t* d_rows, uint64_t cap, uint64_t* n_out);
int bg_oriented_rows(bg_ctx* ctx, const bg_reads* reads, int W_out, uint64_t* d_rows /* 2n x (W_out+1) */);
int bg_rows_prefix_hist(bg_ctx* ctx, const uint64_t* d_rows, uint64_t n, int row_words, int bases, uint64_t every,
                        uint64_t* d_hist);
int bg_cover_build(bg_ctx* ctx, const uint64_t* d_rows, uint64_t n, int W, int cover_k, bg_cover** out);
int bg_cover_test(bg_ctx* ctx, const bg_cover* c, const uint64_t* d_rows, uint64_t n, uint8_t* d_covered);
void bg_cover_free(bg_cover* c);
int bg_cover_query_begin(bg_ctx* ctx, const uint64_t* d_rows, uint64_t n, int W, int cover_k, bg_cover_query** out);
uint64_t bg_cover_query_pending(const bg_cover_query* q);
int bg_cover_query_candidates(bg_ctx* ctx, bg_cover_query* q, int j_lo, int j_hi, uint64_t* d_rows, uint64_t cap,
                              uint64_t* n_out);
/* the two in one (candidates + BG_PART_PREFIX partition): rows are written once, already grouped by destination;
 * h_params = {prefix bases, nparts - 1 ascending splitters}, h_counts[nparts] = rows per destination */
int bg_cover_query_candidates_split(bg_ctx* ctx, bg_cover_query* q, int j_lo, int j_hi, uint32_t nparts,
                                    const uint64_t* h_params, uint64_t* d_out_rows, uint64_t cap, uint64_t* h_counts,
                                    uint64_t* n_out);
int bg_cover_query_answers(bg_ctx* ctx, bg_cover_query* q, const uint64_t* d_rows_sent, const uint8_t* d_covered,
                           uint64_t n, uint8_t* d_keep_out);
int bg_cover_query_plain(bg_ctx* ctx, bg_cover_query* q, uint64_t* d_rows, uint64_t cap, uint64_t* n_out);
void bg_cover_query_free(bg_cover_query* q);
/* Filtered rounds (csrc/shard.hip): a suffix can only be covered by a read that starts with its first K bases, so every
 * owner marks the first K-mers of the reads it indexes in a bit filter of 2^log2_words 64-bit words
 * (bg_cover_filter_build); the owners' filters, all-gathered in rank order (d_filter = nparts x 2^log2_words words), let
 * the sender skip every suffix whose K-mer no read starts with -- a miss is exact, such a suffix is a record.
 *   bg_cover_query_scan          per pending read the next <= max_hits suffixes the owner's filter knows, as candidate rows
 *                                in partitioned order (bg_cover_test them at the owner); a read that runs out is settled
 *   bg_cover_query_scan_answers  answers in sent order: the first verified cover settles a read, the others scan on
 *   bg_cover_query_records_split once nothing is pending: the records (suffixes 1 <= j < first cover, short tails
 *                                included) of the window [j_lo, j_hi), in partitioned order; _records_live = reads that
 *                                have records at or beyond j_lo (x (j_hi - j_lo) bounds the window's rows)
 * Same records as bg_cover_query_candidates / _answers / _plain, with about one candidate per read instead of nine. */
int bg_cover_filter_build(bg_ctx* ctx, const bg_cover* c, int log2_words, uint64_t* d_bits);
int bg_cover_query_scan(bg_ctx* ctx, bg_cover_query* q, const uint64_t* d_filter, int log2_words, uint32_t max_hits, uint32_t nparts,
                        const uint64_t* h_params, uint64_t* d_out_rows, uint64_t cap, uint64_t* h_counts, uint64_t* n_out);
int bg_cover_query_scan_answers(bg_ctx* ctx, bg_cover_query* q, const uint64_t* d_rows_sent, const uint8_t* d_covered, uint64_t n);
int bg_cover_query_records_split(bg_ctx* ctx, bg_cover_query* q, int j_lo, int j_hi, uint32_t nparts, const uint64_t* h_params,
                                 uint64_t* d_out_rows, uint64_t cap, uint64_t* h_counts, uint64_t* n_out);
int bg_cover_query_records_live(bg_ctx* ctx, const bg_cover_query* q, int j_lo, uint64_t* n_out);

/* ---- One device arena for the library AND torch (optional; biograph_amd.use_shared_arena()).  bg_torch_malloc /
 * bg_torch_free have the signature torch.cuda.memory.CUDAPluggableAllocator expects; once torch allocates through
 * them (or after bg_torch_allocator_enable), every context created afterwards takes its memory from the same
 * per-device arena, and bg_ctx_memory reports that arena (library and torch together). */
void bg_torch_allocator_enable(void);
int bg_torch_allocator_enabled(void);
void* bg_torch_malloc(ssize_t size, int device, void* stream);
void bg_torch_free(void* ptr, ssize_t size, int device, void* stream);

/* ---- device memory helpers for hosts without their own allocator (CLI, ctypes tests). */
int bg_dev_alloc(bg_ctx* ctx, uint64_t bytes, void** d_out);
int bg_dev_free(bg_ctx* ctx, void* d_ptr);
int bg_dev_upload(bg_ctx* ctx, void* d_dst, const void* h_src, uint64_t bytes);
int bg_dev_download(bg_ctx* ctx, void* h_dst, const void* d_src, uint64_t bytes);

#ifdef __cplusplus
}
#endif
#endif /* BGSEQ_H_ */
