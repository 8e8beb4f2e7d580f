// bgbinary — command-line front end over libbgseq.so mirroring the reference's `bgbinary create`
// (modules/biograph/biograph_create.cpp:258-335 flags, :516-527 output layout) and the reference-map
// stage of `bgbinary discovery` (modules/biograph/biograph_variants.cpp:642-667).
//
//   bgbinary create --reads X.fq[.gz]|X.bam [--pair Y.fq] [--interleaved] [--ref DIR] --out S.bg [--id ID] [-f]
//                   [--kmer-size 30] [--min-kmer-count 5] [--trim-after-portion 0.7]
//                   [--max-corrections 8] [--min-good-run 2] [--min-reads 0.4] [--warn-reads 0.7] [--stats FILE]
//   bgbinary refmap --in S.bg --ref DIR|FASTA --out FILE          (ref_map::build -> one byte per entry)
//   bgbinary discovery --in S.bg --ref DIR|FASTA --out V.vcf [--sample ID] [--min-overlap 0.7|N]   (graph_discover pipeline)
//   bgbinary merge --in A.bg --in B.bg [...] --out M.bg [--id ID] [-f]   (biograph_merge.cpp:189-400)
//
// Host work here is file I/O only (FASTQ parsing with zlib, output directory); every computation is a
// C-ABI call that runs on the GPU.  The sample's readmap lands in coverage/<sha1>.readmap and is listed
// under "samples" in metadata/bg_info.json (biograph_create.cpp:812-831, biograph_dir.cpp).
#include <dlfcn.h>
#include <errno.h>
#include <malloc.h>
#include <fcntl.h>
#include <stdarg.h>
#include <string.h>
#include <signal.h>
#include <sys/mman.h>
#include <sched.h>
#include <sys/stat.h>
#include <sys/wait.h>
#include <unistd.h>
#include <time.h>
#include <zlib.h>
#if defined(__x86_64__)
#include <immintrin.h>
#endif

#include <algorithm>
#include <array>
#include <atomic>
#include <condition_variable>
#include <deque>
#include <mutex>
#include <thread>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <functional>
#include <map>
#include <memory>
#include <string>
#include <string_view>
#include <unordered_map>
#include <vector>

#include "../../../include/bgseq.h"
#define FASTALLOC_DEFINE
#include "cpuprof.h"    // BG_CPU_PROFILE=FILE: a sampling profile of the command (off unless set)
#include "fastalloc.h"  // thread-caching operator new / delete: the tracer sessions of `discovery` live on small allocations

static void die(const std::string& msg) {
  fprintf(stderr, "%s\n", msg.c_str());
  exit(1);
}
static void check(int rc, const char* what) {
  if (rc != BG_OK) die(std::string(what) + ": " + bg_last_error());
}

struct Args {
  std::map<std::string, std::vector<std::string>> opt;
  bool has(const std::string& k) const { return opt.count(k) != 0; }
  std::string get(const std::string& k, const std::string& dflt = "") const {
    auto it = opt.find(k);
    return it == opt.end() || it->second.empty() ? dflt : it->second.back();
  }
};

// Options every command understands are listed per command: anything else is an error, never silently
// ignored (an interleaved FASTQ imported as unpaired, a --cut-reads that cuts nothing).
static Args parse(int argc, char** argv, int first, const std::vector<std::string>& known) {
  static const char* flags[] = {"interleaved", "force", "f", "allow-long-reads", "no-bwt", "prefault", "no-prefault", "force-sharded", nullptr};
  Args a;
  for (int i = first; i < argc; ++i) {
    std::string s = argv[i];
    if (s.rfind("--", 0) != 0 && s != "-f") die("unexpected argument: " + s);
    std::string key = s == "-f" ? "force" : s.substr(2), val;
    size_t eq = key.find('=');
    if (eq != std::string::npos) { val = key.substr(eq + 1); key = key.substr(0, eq); }
    bool is_flag = false;
    for (const char** f = flags; *f; ++f) is_flag |= key == *f;
    if (eq == std::string::npos && !is_flag) { if (i + 1 >= argc) die("missing value for --" + key); val = argv[++i]; }
    if (key == "in") key = "reads";  // --reads,in (biograph_create.cpp:263)
    if (std::find(known.begin(), known.end(), key) == known.end())
      die("unknown or unsupported option --" + key + " (this build does not implement it; refusing to ignore it)");
    a.opt[key].push_back(val);
  }
  return a;
}

static std::string json_escape(const std::string& s) {
  std::string o;
  for (unsigned char c : s) {
    if (c == '"' || c == '\\') { o.push_back('\\'); o.push_back(char(c)); }
    else if (c == '\n') o += "\\n";
    else if (c == '\t') o += "\\t";
    else if (c < 0x20) { char b[8]; snprintf(b, sizeof b, "\\u%04x", c); o += b; }
    else o.push_back(char(c));
  }
  return o;
}
static std::string json_str(const std::string& s) { return "\"" + json_escape(s) + "\""; }
static std::string json_argv(int argc, char** argv) {
  std::string cmd;
  for (int i = 0; i < argc; ++i) cmd += std::string(i ? " " : "") + argv[i];
  return json_str(cmd);
}

// One input file: plain text stays in the file (its bytes go from the descriptor to the GPU, bg_dev_upload_file), gzip
// and pipes are inflated / read into host memory (zlib; GPU-side gzip is out of scope).
struct HostText {
  int fd = -1;          // >= 0: a plain regular file of n bytes
  size_t n = 0;
  std::string inflated;
};
static void load_text(const std::string& path, HostText* t) {
  int fd = open(path == "-" ? "/dev/stdin" : path.c_str(), O_RDONLY);
  if (fd < 0) die("Could not open " + path + ": " + strerror(errno));
  unsigned char magic[2] = {0, 0};
  struct stat st;
  const bool regular = fstat(fd, &st) == 0 && S_ISREG(st.st_mode);
  if (regular && st.st_size > 0 && pread(fd, magic, 2, 0) == 2 && !(magic[0] == 0x1f && magic[1] == 0x8b)) {
    t->fd = fd;
    t->n = size_t(st.st_size);
    return;
  }
  close(fd);
  gzFile f = gzopen(path == "-" ? "/dev/stdin" : path.c_str(), "rb");  // transparent for plain text (pipes)
  if (!f) die("Could not open " + path + ": " + strerror(errno));
  gzbuffer(f, 1 << 22);
  std::vector<char> buf(1 << 24);
  int got;
  while ((got = gzread(f, buf.data(), unsigned(buf.size()))) > 0) t->inflated.append(buf.data(), size_t(got));
  if (got < 0) die(path + ": read error (truncated gzip?)");
  gzclose(f);
  t->n = t->inflated.size();
}
static void free_text(HostText* t) {
  if (t->fd >= 0) close(t->fd);
  t->fd = -1;
  std::string().swap(t->inflated);
}

// ---- BAM input (read_importer.cpp:182-264,387-470 without htslib).  A BAM file is a BGZF stream, i.e. concatenated
// gzip members, which zlib's gzread inflates as one stream (load_text); the records are decoded here into the FASTQ text
// the GPU importer takes: secondary / supplementary alignments are skipped, a read aligned to the reverse strand is
// turned back (reverse_complement_iupac_string, dna_base_set.cpp:79-94), paired reads are matched by name through a
// cache shared by all input files (the later mate first, as add_paired_read receives them), and paired reads whose mate
// never shows up are treated as unpaired ("entries remaining in pair cache").  CRAM needs the reference-based codecs of
// htslib and is not read.
struct BamState {
  std::unordered_map<std::string, std::string> cache;  // qname -> sequence of the first mate seen
  std::string pairs_text;                              // interleaved FASTQ records of completed pairs
  std::vector<std::pair<std::string, std::string>> unpaired;
  bool got_paired = false;
  uint64_t n_records = 0, n_skipped = 0;
};
static bool is_bam(const std::string& data) { return data.size() >= 4 && memcmp(data.data(), "BAM\1", 4) == 0; }
static void fastq_record(std::string* out, const std::string& name, const std::string& seq) {
  out->push_back('@');
  out->append(name);
  out->push_back('\n');
  out->append(seq);
  out->append("\n+\n");
  out->append(seq.size(), 'I');
  out->push_back('\n');
}
static void iupac_reverse_complement(std::string* s) {
  static const char comp[] = "TVGH..CD..M.KN...YSA.BW.R.";  //  ABCD..GH..K.MN...RST.VW.Y.
  std::reverse(s->begin(), s->end());
  for (char& c : *s) {
    const int idx = int(c) - 'A';
    if (idx >= 0 && idx < int(sizeof comp) - 1) c = comp[idx];
  }
}
static void bam_records(const std::string& data, const std::string& path, BamState* st) {
  auto rd32 = [&](size_t at) -> int32_t {
    if (at + 4 > data.size()) die(path + ": truncated BAM");
    uint32_t v;
    memcpy(&v, data.data() + at, 4);
    return int32_t(v);
  };
  size_t at = 4;
  const int32_t l_text = rd32(at);
  at += 4;
  if (l_text < 0 || at + size_t(l_text) > data.size()) die(path + ": truncated BAM header");
  at += size_t(l_text);
  const int32_t n_ref = rd32(at);
  at += 4;
  for (int32_t i = 0; i < n_ref; ++i) {
    const int32_t l_name = rd32(at);
    if (l_name < 0) die(path + ": corrupt BAM reference list");
    at += 4 + size_t(l_name) + 4;
  }
  if (at > data.size()) die(path + ": truncated BAM reference list");
  static const char nt16[] = "=ACMGRSVTWYHKDBN";
  while (at < data.size()) {
    const int32_t block = rd32(at);
    at += 4;
    if (block < 32 || at + size_t(block) > data.size()) die(path + ": truncated BAM record");
    const unsigned char* r = reinterpret_cast<const unsigned char*>(data.data()) + at;
    const unsigned l_read_name = r[8];
    const unsigned n_cigar = unsigned(r[12]) | unsigned(r[13]) << 8;
    const unsigned flag = unsigned(r[14]) | unsigned(r[15]) << 8;
    int32_t l_seq;
    memcpy(&l_seq, r + 16, 4);
    const size_t need = 32 + size_t(l_read_name) + 4 * size_t(n_cigar) + (size_t(l_seq < 0 ? 0 : l_seq) + 1) / 2 + size_t(l_seq < 0 ? 0 : l_seq);
    if (l_seq < 0 || need > size_t(block) || !l_read_name) die(path + ": corrupt BAM record");
    at += size_t(block);
    ++st->n_records;
    if (flag & (0x100 | 0x800)) { ++st->n_skipped; continue; }  // BAM_FSECONDARY | BAM_FSUPPLEMENTARY
    std::string qname(reinterpret_cast<const char*>(r + 32), l_read_name - 1);
    const unsigned char* sq = r + 32 + l_read_name + 4 * size_t(n_cigar);
    std::string seq(size_t(l_seq), 'N');
    for (int32_t i = 0; i < l_seq; ++i) seq[size_t(i)] = nt16[(sq[i >> 1] >> ((~i & 1) << 2)) & 15];
    if (flag & 0x10) iupac_reverse_complement(&seq);  // BAM_FREVERSE
    if (flag & 0x1) {                                 // BAM_FPAIRED
      st->got_paired = true;
      auto it = st->cache.find(qname);
      if (it != st->cache.end()) {
        fastq_record(&st->pairs_text, qname, seq);
        fastq_record(&st->pairs_text, qname, it->second);
        st->cache.erase(it);
      } else {
        st->cache.emplace(std::move(qname), std::move(seq));
      }
    } else {
      st->unpaired.emplace_back(std::move(qname), std::move(seq));
    }
  }
}
// -> the text for bg_fastq_import: interleaved pairs (an unpaired read = a pair whose mate is an empty record) when any
// read was flagged paired, else plain records
static std::string bam_finish(BamState* st, uint64_t* n_lacking) {
  *n_lacking = st->cache.size();
  std::vector<std::pair<std::string, std::string>> rest(st->cache.begin(), st->cache.end());
  std::sort(rest.begin(), rest.end());  // (the reference walks hash buckets here; any order gives the same biograph)
  st->cache.clear();
  std::string text = std::move(st->pairs_text);
  for (auto* lst : {&st->unpaired, &rest})
    for (auto& u : *lst) {
      fastq_record(&text, u.first, u.second);
      if (st->got_paired) fastq_record(&text, u.first, std::string());
    }
  return text;
}

static bool exists(const std::string& p) { struct stat st; return stat(p.c_str(), &st) == 0; }
static void mkdirs(const std::string& p) {
  if (mkdir(p.c_str(), 0777) != 0 && errno != EEXIST) die("Could not create " + p + ": " + strerror(errno));
}
static void write_file(const std::string& p, const std::string& data) {
  FILE* f = fopen(p.c_str(), "wb");
  if (!f || fwrite(data.data(), 1, data.size(), f) != data.size()) die("Could not write " + p);
  fclose(f);
}

static double now_s() {
  timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return ts.tv_sec + ts.tv_nsec * 1e-9;
}

// resident set of the process in GB (/proc/self/statm), for the timing lines of `discovery`
static double rss_gb() {
  FILE* f = fopen("/proc/self/statm", "r");
  if (!f) return 0;
  unsigned long long total = 0, res = 0;
  const int n = fscanf(f, "%llu %llu", &total, &res);
  fclose(f);
  return n == 2 ? double(res) * double(sysconf(_SC_PAGESIZE)) / 1e9 : 0;
}
// seconds since the process started (/proc/self/stat field 22 against the uptime): what the loader and the runtimes' static
// initialisers took before main
static double since_process_start_s() {
  FILE* f = fopen("/proc/self/stat", "r");
  if (!f) return 0;
  char buf[2048];
  const size_t n = fread(buf, 1, sizeof buf - 1, f);
  fclose(f);
  buf[n] = 0;
  const char* p = strrchr(buf, ')');
  if (!p) return 0;
  unsigned long long start = 0;
  int field = 2;
  for (p = p + 1; *p && field < 22; ++p)
    if (*p == ' ') { ++field; if (field == 22) { start = strtoull(p + 1, nullptr, 10); break; } }
  double up = 0;
  FILE* u = fopen("/proc/uptime", "r");
  if (u) { if (fscanf(u, "%lf", &up) != 1) up = 0; fclose(u); }
  return up - double(start) / double(sysconf(_SC_CLK_TCK));
}

static std::string g_log;  // qc/create_log.txt (biograph_create.cpp:527): what went to stderr, with times
static void logf(const char* fmt, ...) {
  char buf[1024];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  fprintf(stderr, "%s\n", buf);
  time_t now = time(nullptr);
  char tb[32];
  struct tm tmv;
  localtime_r(&now, &tmv);
  strftime(tb, sizeof tb, "%Y-%m-%d %H:%M:%S", &tmv);
  g_log += std::string(tb) + " " + buf + "\n";
}

static std::string stem_of(std::string p) {  // accession id default: stem of the output directory (:518-520)
  while (p.size() > 1 && p.back() == '/') p.pop_back();
  size_t sl = p.rfind('/');
  if (sl != std::string::npos) p = p.substr(sl + 1);
  size_t dot = p.rfind('.');
  return dot == std::string::npos || dot == 0 ? p : p.substr(0, dot);
}

// bgbinary bam2fastq --reads X.bam [--reads Y.bam] --out R.fq: the text `create` hands to the GPU importer for BAM input
// (interleaved when the BAM holds paired reads).  Host only; also what the CPU tests check the decoder with.
static int cmd_bam2fastq(int argc, char** argv) {
  Args a = parse(argc, argv, 2, {"reads", "out"});
  if (!a.has("reads") || !a.has("out")) die("bam2fastq: --reads and --out are required");
  BamState bam;
  for (auto& f : a.opt["reads"]) {
    HostText t;
    load_text(f, &t);
    if (t.fd >= 0 || !is_bam(t.inflated)) die(f + ": not a BAM file");
    bam_records(t.inflated, f, &bam);
    free_text(&t);
  }
  uint64_t lacking = 0;
  const bool paired = bam.got_paired;
  const uint64_t n_rec = bam.n_records, n_skip = bam.n_skipped;
  write_file(a.get("out"), bam_finish(&bam, &lacking));
  fprintf(stderr, "%llu records, %llu skipped, %llu without mate, %s\n", (unsigned long long)n_rec, (unsigned long long)n_skip,
          (unsigned long long)lacking, paired ? "paired (interleaved)" : "unpaired");
  return 0;
}


// ---- `create --gpus N`: one child process per GPU, forked before anything touches HIP (biograph_create.cpp:431-811 with the
// partitions of part_repo.cpp:32-45 owned by ranks instead of files).  Every rank imports its own stretch of the input, the
// ranks build ONE sample together (bg_create_sharded over RCCL), write one seqset container in place and rank 0 writes the
// readmap and the metadata.

// newlines in [lo, hi) of a source (a plain file through pread, or host text), on a few threads
struct TextSrc {
  int fd = -1;
  const char* mem = nullptr;
  uint64_t n = 0;
};
static uint64_t count_newlines_block(const TextSrc& t, uint64_t lo, uint64_t hi, std::vector<char>* buf) {
  uint64_t c = 0;
  if (t.mem) {
    const char* p = t.mem + lo;
    const char* e = t.mem + hi;
    while ((p = static_cast<const char*>(memchr(p, '\n', size_t(e - p))))) { ++c; ++p; }
    return c;
  }
  buf->resize(size_t(4) << 20);
  for (uint64_t at = lo; at < hi;) {
    const size_t want = size_t(std::min<uint64_t>(buf->size(), hi - at));
    ssize_t got = pread(t.fd, buf->data(), want, off_t(at));
    if (got <= 0) die("read error while splitting the input over the ranks");
    const char* p = buf->data();
    const char* e = p + got;
    while ((p = static_cast<const char*>(memchr(p, '\n', size_t(e - p))))) { ++c; ++p; }
    at += uint64_t(got);
  }
  return c;
}
static uint64_t count_newlines(const TextSrc& t, uint64_t lo, uint64_t hi, int threads) {
  if (hi <= lo) return 0;
  const int T = int(std::max<uint64_t>(1, std::min<uint64_t>(uint64_t(threads > 0 ? threads : 8), (hi - lo) >> 22)));
  std::vector<uint64_t> part(T, 0);
  std::vector<std::thread> th;
  for (int i = 0; i < T; ++i)
    th.emplace_back([&, i] {
      std::vector<char> buf;
      part[i] = count_newlines_block(t, lo + (hi - lo) * uint64_t(i) / T, lo + (hi - lo) * uint64_t(i + 1) / T, &buf);
    });
  for (auto& x : th) x.join();
  uint64_t c = 0;
  for (uint64_t x : part) c += x;
  return c;
}
// byte offset just behind the k-th newline (k >= 1) at or after `lo`
static uint64_t after_kth_newline(const TextSrc& t, uint64_t lo, uint64_t k) {
  std::vector<char> buf(size_t(4) << 20);
  for (uint64_t at = lo; at < t.n;) {
    const size_t want = size_t(std::min<uint64_t>(buf.size(), t.n - at));
    const char* base;
    ssize_t got;
    if (t.mem) { base = t.mem + at; got = ssize_t(want); }
    else { got = pread(t.fd, buf.data(), want, off_t(at)); base = buf.data(); if (got <= 0) die("read error while splitting the input over the ranks"); }
    const char* p = base;
    const char* e = base + got;
    while ((p = static_cast<const char*>(memchr(p, '\n', size_t(e - p))))) {
      ++p;
      if (--k == 0) return at + uint64_t(p - base);
    }
    at += uint64_t(got);
  }
  return t.n;
}
// The stretch [*lo, *hi) of one source that holds this rank's records.  Records are four lines; `unit` records stay together
// (2 for an interleaved file); `n_records` (in/out): 0 = derive from this source, else the count the sources of a pair share.
// Every rank scans only its own n / world bytes, the counts meet through the communicator.
static void my_stretch(const TextSrc& t, int rank, int world, bg_comm* comm, uint64_t unit, int threads, uint64_t* n_records, uint64_t* lo, uint64_t* hi) {
  std::vector<uint64_t> cnt(world, 0);
  const uint64_t a = t.n * uint64_t(rank) / world, b = t.n * uint64_t(rank + 1) / world;
  uint64_t mine[2] = {count_newlines(t, a, b, threads), 0};
  if (rank == world - 1 && t.n) {  // a last line without its newline still is a line
    char last = 0;
    if (t.mem) last = t.mem[t.n - 1]; else if (pread(t.fd, &last, 1, off_t(t.n - 1)) != 1) die("read error");
    mine[1] = last != '\n';
  }
  std::vector<uint64_t> all(size_t(world) * 2);
  check(bg_comm_all_gather_host(comm, mine, sizeof mine, all.data()), "bg_comm_all_gather_host");
  uint64_t lines = 0;
  std::vector<uint64_t> before(world + 1, 0);
  for (int q = 0; q < world; ++q) { before[q + 1] = before[q] + all[size_t(q) * 2]; lines += all[size_t(q) * 2] + all[size_t(q) * 2 + 1]; }
  if (!*n_records) *n_records = lines / 4;
  const uint64_t units = *n_records / unit;
  const uint64_t first_rec = (units * uint64_t(rank) / world) * unit;
  uint64_t start = 0;
  if (first_rec) {  // behind newline number 4 * first_rec, which lies in the stretch of some rank q
    const uint64_t k = 4 * first_rec;
    int q = 0;
    while (q + 1 < world && before[q + 1] < k) ++q;
    start = after_kth_newline(t, t.n * uint64_t(q) / world, k - before[q]);
  }
  std::vector<uint64_t> starts(world);
  check(bg_comm_all_gather_host(comm, &start, 8, starts.data()), "bg_comm_all_gather_host");
  *lo = start;
  *hi = rank + 1 < world ? starts[rank + 1] : t.n;  // the last rank takes what is left (blank tail lines, a mate file that is longer)
}

// a rank of `--gpus N`: its device, its context, the communicator (rank 0 draws the RCCL id, the launcher's pipes carry it on)
static void rank_open(int rank, int world, int device, const std::string& backend, const char* comm_id, int id_fd_in,
                      const std::vector<int>& id_fd_out, bg_ctx** ctx, bg_comm** comm) {
  int ndev = bg_device_count();
  if (ndev < 1) die("--gpus: no HIP device");
  if (device < 0) device = backend == "shm" ? rank % ndev : rank;
  if (device >= ndev) die("--gpus " + std::to_string(world) + ": rank " + std::to_string(rank) + " has no device " + std::to_string(device) +
                          " (" + std::to_string(ndev) + " visible; --comm shm lets ranks share a device)");
  check(bg_ctx_create(device, nullptr, ctx), "bg_ctx_create");
  char id[BG_COMM_ID_BYTES];
  memset(id, 0, sizeof id);
  if (backend == "rccl") {
    if (rank == 0) {
      check(bg_comm_unique_id(id), "bg_comm_unique_id");
      for (int fd : id_fd_out)
        if (write(fd, id, sizeof id) != ssize_t(sizeof id)) die("--gpus: could not hand the communicator id on");
    } else if (read(id_fd_in, id, sizeof id) != ssize_t(sizeof id)) {
      die("--gpus: rank 0 never sent the communicator id");
    }
  } else {
    snprintf(id, sizeof id, "%s", comm_id);
  }
  check(bg_comm_init(*ctx, rank, world, id, backend.c_str(), comm), "bg_comm_init");
}

static int create_rank(const Args& a, int argc, char** argv, const bg_create_params& p, const std::string& out, const std::vector<std::string>& files,
                       bool paired_files, bool interleaved, const std::string& fmt, uint32_t cut_start, uint32_t cut_end, int threads, int rank, int world,
                       int device, const std::string& backend, const char* comm_id, int id_fd_in, const std::vector<int>& id_fd_out, double min_reads,
                       double warn_reads) {
  const bool lead = rank == 0;
  double t0 = now_s();
  bg_ctx* ctx = nullptr;
  bg_comm* comm = nullptr;
  rank_open(rank, world, device, backend, comm_id, id_fd_in, id_fd_out, &ctx, &comm);
  if (lead) logf("Importing reads on %d rank(s) (communicator: %s)", world, backend.c_str());

  // ---- this rank's stretch of every input
  std::vector<HostText> texts(files.size());
  std::vector<TextSrc> srcs(files.size());
  BamState bam;
  bool any_bam = false;
  std::string bam_text;
  for (size_t i = 0; i < files.size(); ++i) {
    load_text(files[i], &texts[i]);
    const bool this_bam = texts[i].fd < 0 && is_bam(texts[i].inflated);
    if (fmt == "bam" && !this_bam) die(files[i] + ": not a BAM file");
    if (this_bam) {
      if (fmt == "fastq") die(files[i] + ": a BAM file was given with --format fastq");
      if (paired_files || interleaved) die("--pair / --interleaved do not apply to BAM input (mates are matched by name)");
      if (i && !any_bam) die("BAM and FASTQ inputs cannot be mixed in one run");
      any_bam = true;
      bam_records(texts[i].inflated, files[i], &bam);  // (every rank decodes the whole file: a BGZF stream has no record index)
      free_text(&texts[i]);
      continue;
    }
    if (any_bam) die("BAM and FASTQ inputs cannot be mixed in one run");
    srcs[i].fd = texts[i].fd;
    srcs[i].mem = texts[i].fd >= 0 ? nullptr : texts[i].inflated.data();
    srcs[i].n = texts[i].n;
  }
  bool bam_paired = false;
  if (any_bam) {
    uint64_t lacking = 0;
    bam_text = bam_finish(&bam, &lacking);
    bam_paired = bam.got_paired;
    if (lead) logf("BAM: %llu records (%llu secondary / supplementary skipped)%s", (unsigned long long)bam.n_records, (unsigned long long)bam.n_skipped,
                   bam_paired ? ", paired" : ", unpaired");
    if (lead && lacking) logf("WARNING: %llu entries remaining in pair cache from paired reads; treating as unpaired reads", (unsigned long long)lacking);
    srcs.assign(1, TextSrc());
    srcs[0].mem = bam_text.data();
    srcs[0].n = bam_text.size();
    if (bam_paired && lacking) die("create --gpus: a BAM whose pairs are incomplete cannot be split over ranks (paired and unpaired records alternate)");
  }
  const bool inter = interleaved || bam_paired;
  uint64_t pair_records = 0;
  std::vector<void*> d_text(srcs.size(), nullptr);
  std::vector<uint64_t> n_text(srcs.size(), 0);
  uint64_t text_bytes = 0;
  for (size_t i = 0; i < srcs.size(); ++i) {
    uint64_t lo = 0, hi = srcs[i].n;
    if (world > 1) {
      uint64_t n_rec = 0;
      if (paired_files && (i & 1)) n_rec = pair_records;  // the mate file follows the record count of its partner
      my_stretch(srcs[i], rank, world, comm, inter ? 2 : 1, threads, &n_rec, &lo, &hi);
      if (paired_files && !(i & 1)) pair_records = n_rec;
    }
    n_text[i] = hi - lo;
    text_bytes += n_text[i];
    check(bg_dev_alloc(ctx, n_text[i] ? n_text[i] : 1, &d_text[i]), "bg_dev_alloc");
    if (srcs[i].mem) check(bg_dev_upload_pinned(ctx, d_text[i], srcs[i].mem + lo, n_text[i], threads), "upload");
    else check(bg_dev_upload_file(ctx, d_text[i], srcs[i].fd, lo, n_text[i], threads), "upload");
  }
  for (auto& t : texts) free_text(&t);
  std::string().swap(bam_text);
  uint8_t* d_seq = nullptr;
  uint64_t* d_offs = nullptr;
  uint64_t n_reads = 0, n_bases = 0;
  std::vector<const uint8_t*> tp;
  for (void* q : d_text) tp.push_back(static_cast<const uint8_t*>(q));
  check(bg_fastq_import(ctx, uint32_t(tp.size()), tp.data(), n_text.data(), paired_files ? 1 : 0, inter ? 1 : 0, cut_start, cut_end, &d_seq, &d_offs, &n_reads,
                        &n_bases),
        "import");
  for (void* q : d_text) bg_dev_free(ctx, q);
  uint64_t mine3[3] = {n_reads, n_bases, text_bytes}, tot3[3] = {0, 0, 0};
  std::vector<uint64_t> all3(size_t(world) * 3);
  check(bg_comm_all_gather_host(comm, mine3, sizeof mine3, all3.data()), "bg_comm_all_gather_host");
  for (int q = 0; q < world; ++q)
    for (int j = 0; j < 3; ++j) tot3[j] += all3[size_t(q) * 3 + j];
  if (!tot3[0]) die("\nNo reads were imported, exiting.");
  double t_import = now_s() - t0;
  if (lead) logf("Total reads imported: %llu (%llu bases, %.2f GB of FASTQ; %.3f s on %d rank(s))", (unsigned long long)tot3[0], (unsigned long long)tot3[1],
                 tot3[2] / 1e9, t_import, world);

  if (lead) logf("Running kmerization, read correction, seqset and readmap build on %d GPU rank(s)", world);
  double t1 = now_s();
  const int is_paired = (paired_files || inter) ? 1 : 0;
  bg_sharded* sh = nullptr;
  bg_sharded_stats st;
  check(bg_create_sharded(ctx, comm, d_seq, d_offs, n_reads, &p, is_paired ? 2 : 1, &sh, &st), "bg_create_sharded");
  double t_build = now_s() - t1;
  bg_dev_free(ctx, d_seq);
  bg_dev_free(ctx, d_offs);
  const uint64_t all_reads = tot3[0];
  double pct = double(st.build.n_corrected_reads) / double(all_reads);
  char msg[320];
  if (pct < min_reads) {  // biograph_create.cpp:748-757 (every rank sees the same totals)
    snprintf(msg, sizeof msg, "Fewer than %2.0f%% of reads (set by --min-reads) were kept after correction (%llu / %llu remain). Cannot continue.",
             min_reads * 100.0, (unsigned long long)st.build.n_corrected_reads, (unsigned long long)all_reads);
    if (lead) die(msg);
    exit(1);
  }
  if (lead && pct < warn_reads)
    logf("Warning: Fewer than %2.0f%% of reads (set by --warn-reads) survived correction (%llu / %llu remain)", warn_reads * 100.0,
         (unsigned long long)st.build.n_corrected_reads, (unsigned long long)all_reads);
  if (lead) {
    mkdirs(out);
    mkdirs(out + "/qc");
    mkdirs(out + "/metadata");
    mkdirs(out + "/coverage");
  }
  double t2 = now_s();
  std::vector<const char*> av(argv, argv + argc);
  char uuid[40] = "", sha[44] = "";
  const std::string rm_tmp = out + "/coverage/tmp.readmap";
  check(bg_sharded_write(ctx, comm, sh, (out + "/seqset").c_str(), rm_tmp.c_str(), argc, av.data(), uuid, sha), "bg_sharded_write");
  bg_comm_stats cs;
  check(bg_comm_get_stats(comm, &cs), "bg_comm_get_stats");
  uint64_t cs_mine[2] = {cs.bytes_through_backend, cs.collectives};
  std::vector<uint64_t> cs_all(size_t(world) * 2);
  check(bg_comm_all_gather_host(comm, cs_mine, sizeof cs_mine, cs_all.data()), "bg_comm_all_gather_host");
  if (lead) {
    if (rename(rm_tmp.c_str(), (out + "/coverage/" + sha + ".readmap").c_str()) != 0) die("Could not rename " + rm_tmp);
    double t_write = now_s() - t2, t_total = now_s() - t0;
    uint64_t through = 0;
    for (int q = 0; q < world; ++q) through += cs_all[size_t(q) * 2];
    logf("Written: seqset in place by %d rank(s), readmap + digest by rank 0: %.3f s", world, t_write);
    logf("Communicator: %s%s, %llu collectives per rank, %.3f GB through it over all ranks", backend.c_str(),
         cs.backend == 2 ? " (a rank alone: collectives are copies; BG_COMM_NO_SHORTCUT=1 sends them through the backend)" : "",
         (unsigned long long)cs.collectives, through / 1e9);
    std::string id_s = a.get("id");
    if (id_s.empty()) id_s = stem_of(out);
    char nums[1400];
    snprintf(nums, sizeof nums,
             "\"imported_reads\":%llu,\"imported_bases\":%llu,\"corrected_reads\":%llu,\"corrected_bases\":%llu,"
             "\"avg_bases_per_read\":%.6f,\"corrected_pct\":%.6f,\"kept_kmers\":%llu,\"seqset_entries\":%llu,"
             "\"gpus\":%d,\"comm\":\"%s\",\"comm_backend_code\":%d,\"rccl_version\":%d,\"comm_collectives_per_rank\":%llu,\"comm_bytes_through_backend\":%llu,"
             "\"correct_steps\":%u,\"correct_probes\":%llu,\"tails_sent_in_full\":%llu,"
             "\"timings\":[{\"import\":%.3f},{\"gpu_build\":%.3f},{\"write\":%.3f},{\"total\":%.3f}]",
             (unsigned long long)all_reads, (unsigned long long)tot3[1], (unsigned long long)st.build.n_corrected_reads,
             (unsigned long long)st.build.n_corrected_bases,
             st.build.n_corrected_reads ? double(st.build.n_corrected_bases) / double(st.build.n_corrected_reads) : 0.0, pct,
             (unsigned long long)st.build.n_kept_kmers, (unsigned long long)st.build.n_entries, world, backend.c_str(), cs.backend, cs.rccl_version,
             (unsigned long long)cs.collectives, (unsigned long long)through, st.correct_steps, (unsigned long long)st.correct_probes,
             (unsigned long long)st.tails_sent_in_full, t_import, t_build, t_write, t_total);
    std::string js = "{\"command\":\"create\",\"version\":\"biograph_amd-r4\",\"accession_id\":" + json_str(id_s) + ",\"uuid\":" + json_str(uuid) + "," + nums + "}";
    write_file(out + "/qc/create_stats.json", js);
    if (a.has("stats")) write_file(a.get("stats"), js);
    std::string meta = "{\"accession_id\":" + json_str(id_s) + ",\"biograph_id\":" + json_str(uuid) + ",\"command_history\":[" + json_argv(argc, argv) +
                       "],\"samples\":{" + json_str(id_s) + ":" + json_str(sha) + "},\"version\":\"biograph_amd-r4\"}";
    write_file(out + "/metadata/bg_info.json", meta);
    logf("%llu / %llu reads survived read correction; %llu seqset entries; GPU build %.3f s on %d rank(s); %.3f s in all",
         (unsigned long long)st.build.n_corrected_reads, (unsigned long long)all_reads, (unsigned long long)st.build.n_entries, t_build, world, t_total);
    write_file(out + "/qc/create_log.txt", g_log);
  }
  check(bg_comm_barrier(comm), "bg_comm_barrier");
  bg_sharded_free(sh);
  bg_comm_free(comm);
  bg_ctx_destroy(ctx);
  return 0;
}

// One child process per rank, forked before anything touches HIP (a process that has initialised the GPU must not fork ranks
// or exec); body(rank, device, backend, comm_id, id pipe in, id pipes out) runs in the child.  A rank that dies ends the
// command: the others would wait for it forever.
static int launch_ranks(const Args& a, int gpus,
                        const std::function<int(int, int, const std::string&, const char*, int, const std::vector<int>&)>& body) {
  const std::string backend = a.get("comm", "rccl");
  if (backend != "rccl" && backend != "shm") die("--comm takes rccl or shm");
  std::vector<int> devices;
  if (a.has("devices")) {
    const std::string v = a.get("devices");
    for (size_t at = 0; at < v.size();) {
      size_t e = v.find(',', at);
      if (e == std::string::npos) e = v.size();
      devices.push_back(atoi(v.substr(at, e - at).c_str()));
      at = e + 1;
    }
    if (int(devices.size()) != gpus) die("--devices names one device per rank");
  }
  char comm_id[64];
  snprintf(comm_id, sizeof comm_id, "%ld-%ld", long(getpid()), long(time(nullptr)));
  std::vector<std::array<int, 2>> pipes(gpus > 1 ? gpus - 1 : 0);
  for (auto& pp : pipes)
    if (pipe(pp.data()) != 0) die(std::string("pipe: ") + strerror(errno));
  fflush(stdout);
  fflush(stderr);
  std::vector<pid_t> kids;
  for (int r = 0; r < gpus; ++r) {
    pid_t pid = fork();
    if (pid < 0) die(std::string("fork: ") + strerror(errno));
    if (pid == 0) {
      std::vector<int> outs;
      int in = -1;
      for (int q = 1; q < gpus; ++q) {
        if (r == 0) { outs.push_back(pipes[q - 1][1]); close(pipes[q - 1][0]); }
        else if (q == r) { in = pipes[q - 1][0]; close(pipes[q - 1][1]); }
        else { close(pipes[q - 1][0]); close(pipes[q - 1][1]); }
      }
      int rc = body(r, devices.empty() ? -1 : devices[r], backend, comm_id, in, outs);
      fflush(stdout);
      fflush(stderr);
      _exit(rc);
    }
    kids.push_back(pid);
  }
  for (auto& pp : pipes) { close(pp[0]); close(pp[1]); }
  int failed = 0;
  for (size_t left = kids.size(); left;) {
    int status = 0;
    pid_t pid = wait(&status);
    if (pid < 0) break;
    --left;
    const bool ok = WIFEXITED(status) && WEXITSTATUS(status) == 0;
    if (!ok && !failed) {
      failed = 1;
      for (pid_t k : kids)
        if (k != pid) kill(k, SIGTERM);
    }
  }
  if (backend == "shm") {  // ranks that were ended never unlinked their segments
    shm_unlink((std::string("/bgcomm-") + comm_id).c_str());
    for (int r = 0; r < gpus; ++r) shm_unlink((std::string("/bgcomm-") + comm_id + "-r" + std::to_string(r)).c_str());
  }
  return failed;
}

static int create_sharded(const Args& a, int argc, char** argv, const bg_create_params& p, const std::string& out, const std::vector<std::string>& files,
                          bool paired_files, bool interleaved, const std::string& fmt, uint32_t cut_start, uint32_t cut_end, int threads, int gpus,
                          double min_reads, double warn_reads) {
  for (const auto& f : files)
    if (f == "-") die("create --gpus: every rank reads its own stretch of the input: it cannot be a pipe");
  return launch_ranks(a, gpus, [&](int r, int device, const std::string& backend, const char* comm_id, int in, const std::vector<int>& outs) {
    return create_rank(a, argc, argv, p, out, files, paired_files, interleaved, fmt, cut_start, cut_end, threads, r, gpus, device, backend, comm_id, in,
                       outs, min_reads, warn_reads);
  });
}

static int cmd_create(int argc, char** argv) {
  Args a = parse(argc, argv, 2, {"prefault", "no-prefault", "reads", "pair", "interleaved", "ref", "out", "id", "force", "kmer-size", "min-kmer-count",
                                 "trim-after-portion", "max-corrections", "min-good-run", "min-reads", "warn-reads", "stats",
                                 "cut-reads", "threads", "tmp", "max-mem", "format", "allow-long-reads", "gpus", "force-sharded", "comm", "devices"});
  if (!a.has("reads") || !a.has("out")) die("create: --reads and --out are required");
  std::string out = a.get("out");
  if (out.size() < 3 || out.substr(out.size() - 3) != ".bg") out += ".bg";  // biograph_dir appends .bg
  if (exists(out + "/seqset") && !a.has("force")) die("Refusing to overwrite '" + out + "'. Use --force to override.");
  const std::string fmt = a.get("format", "auto");
  if (fmt != "auto" && fmt != "fastq" && fmt != "bam")
    die("Invalid input format '" + fmt + "' (this build reads FASTQ, plain or gzip / BGZF, and BAM; CRAM needs htslib)");
  bg_create_params p;
  bg_create_params_default(&p);
  p.kmer_size = atoi(a.get("kmer-size", "30").c_str());
  p.min_kmer_count = uint32_t(atoi(a.get("min-kmer-count", "5").c_str()));
  p.correct.trim_after_portion = strtof(a.get("trim-after-portion", "0.7").c_str(), nullptr);
  p.correct.max_corrections = uint32_t(atoi(a.get("max-corrections", "8").c_str()));
  p.correct.min_good_run = uint32_t(atoi(a.get("min-good-run", "2").c_str()));
  double min_reads = atof(a.get("min-reads", "0.4").c_str()), warn_reads = atof(a.get("warn-reads", "0.7").c_str());
  if (p.min_kmer_count < 1) die("--min-kmer-count must be at least 1");
  uint32_t cut_start = 0, cut_end = 0;
  if (a.has("cut-reads") && !a.get("cut-reads").empty()) {  // validate_cut_param, biograph_create.cpp:376-420
    const std::string v = a.get("cut-reads");
    size_t dash = v.find('-');
    if (dash == std::string::npos) die("cut-reads must specify a range separated by a dash");
    char* e1 = nullptr;
    char* e2 = nullptr;
    unsigned long s0 = strtoul(v.substr(0, dash).c_str(), &e1, 10), s1 = strtoul(v.substr(dash + 1).c_str(), &e2, 10);
    if (dash == 0 || dash + 1 == v.size() || *e1 || *e2) die("cut-reads must specify a numerical range; couldn't parse " + v);
    if (s1 <= s0) die("cut-reads: the end of the range must be greater than its start");
    cut_start = uint32_t(s0);
    cut_end = uint32_t(s1);
  }
  const auto& ins = a.opt["reads"];
  const auto& pairs = a.has("pair") ? a.opt["pair"] : std::vector<std::string>();
  const bool interleaved = a.has("interleaved");
  if (!pairs.empty() && pairs.size() != ins.size()) die("If pair files are present, there must be the same number of them as read files.");
  if (interleaved && !pairs.empty()) die("--interleaved and --pair exclude each other");
  const int threads = atoi(a.get("threads", "0").c_str());
  const int gpus = atoi(a.get("gpus", "1").c_str());
  if (gpus < 1 || gpus > 127) die("--gpus takes 1..127");
  if (gpus > 1 || a.has("force-sharded")) {
    std::vector<std::string> files;
    for (size_t i = 0; i < ins.size(); ++i) { files.push_back(ins[i]); if (!pairs.empty()) files.push_back(pairs[i]); }
    return create_sharded(a, argc, argv, p, out, files, !pairs.empty(), interleaved, fmt, cut_start, cut_end, threads, gpus, min_reads, warn_reads);
  }

  double t0 = now_s();
  logf("Importing reads");
  bg_ctx* ctx = nullptr;
  check(bg_ctx_create(0, nullptr, &ctx), "bg_ctx_create");
  // the bytes of every file go to HBM as they are; records are found and paired there (csrc/fastq.hip)
  std::vector<std::string> files;
  for (size_t i = 0; i < ins.size(); ++i) { files.push_back(ins[i]); if (!pairs.empty()) files.push_back(pairs[i]); }
  std::vector<void*> d_text(files.size(), nullptr);
  std::vector<uint64_t> n_text(files.size(), 0);
  uint64_t text_bytes = 0;
  // While the files travel to the GPU, a side thread reserves the memory the build will ask for (bg_ctx_reserve).  The build
  // peaks near 14 bytes per input base; FASTQ text is a little over two bytes per base.
  std::thread prefault;
  // On by default since round 5 (--no-prefault / BG_CREATE_RESERVE=off: without): the reservation maps 1 GiB pieces through the
  // virtual-memory API (0.05 s for 32 GiB; the 256 MiB pieces of round 4 were the slow way), and what the build would
  // otherwise pay inside its first step -- up to 0.95 s when another process has just given its memory back -- happens
  // beside the import.  BG_CREATE_RESERVE=sync: before the import, on this thread (measurements).
  const char* reserve_env = getenv("BG_CREATE_RESERVE");
  const bool reserve_sync = reserve_env && !strcmp(reserve_env, "sync");
  const bool reserve_off = a.has("no-prefault") || (reserve_env && !strcmp(reserve_env, "off"));
  if (!reserve_off) {
    uint64_t on_disk = 0;
    for (auto& f : files) { struct stat sb; if (stat(f.c_str(), &sb) == 0) on_disk += uint64_t(sb.st_size); }
    const bool packed = !files.empty() && (files[0].size() > 3 && (files[0].substr(files[0].size() - 3) == ".gz" || files[0].substr(files[0].size() - 4) == ".bam"));
    const uint64_t est_bases = packed ? on_disk * 2 : on_disk / 2;
    uint64_t want = est_bases * 14;
    if (want > (uint64_t(64) << 30)) want = uint64_t(64) << 30;  // stays inside what the arena may keep cached (35 % of the device)
    if (reserve_sync) {
      const double tr = now_s();
      (void)bg_ctx_reserve(ctx, want);
      logf("Reserved %.1f GB of device memory in %.3f s", want / 1e9, now_s() - tr);
    } else if (want >= (uint64_t(256) << 20)) {
      prefault = std::thread([ctx, want] { (void)bg_ctx_reserve(ctx, want); });
    }
  }
  BamState bam;
  bool any_bam = false;
  std::string bam_text;
  for (size_t i = 0; i < files.size(); ++i) {
    HostText t;
    load_text(files[i], &t);
    const bool this_bam = t.fd < 0 && is_bam(t.inflated);
    if (fmt == "bam" && !this_bam) die(files[i] + ": not a BAM file");
    if (this_bam) {
      if (fmt == "fastq") die(files[i] + ": a BAM file was given with --format fastq");
      if (!pairs.empty() || interleaved) die("--pair / --interleaved do not apply to BAM input (mates are matched by name)");
      if (i && !any_bam) die("BAM and FASTQ inputs cannot be mixed in one run");
      any_bam = true;
      bam_records(t.inflated, files[i], &bam);
      free_text(&t);
      if (i + 1 < files.size()) continue;
      uint64_t lacking = 0;
      bam_text = bam_finish(&bam, &lacking);
      logf("BAM: %llu records (%llu secondary / supplementary skipped)%s", (unsigned long long)bam.n_records,
           (unsigned long long)bam.n_skipped, bam.got_paired ? ", paired" : ", unpaired");
      if (lacking) logf("WARNING: %llu entries remaining in pair cache from paired reads; treating as unpaired reads", (unsigned long long)lacking);
      // one source holding everything
      d_text.assign(1, nullptr);
      n_text.assign(1, bam_text.size());
      text_bytes = bam_text.size();
      check(bg_dev_alloc(ctx, bam_text.size() ? bam_text.size() : 1, &d_text[0]), "bg_dev_alloc");
      check(bg_dev_upload_pinned(ctx, d_text[0], bam_text.data(), bam_text.size(), threads), "upload");
      std::string().swap(bam_text);
      break;
    }
    if (any_bam) die("BAM and FASTQ inputs cannot be mixed in one run");
    n_text[i] = t.n;
    text_bytes += t.n;
    check(bg_dev_alloc(ctx, t.n ? t.n : 1, &d_text[i]), "bg_dev_alloc");
    if (t.fd >= 0) check(bg_dev_upload_file(ctx, d_text[i], t.fd, 0, t.n, threads), "upload");
    else check(bg_dev_upload_pinned(ctx, d_text[i], t.inflated.data(), t.n, threads), "upload");
    free_text(&t);
  }
  double t_upload = now_s() - t0;
  uint8_t* d_seq = nullptr;
  uint64_t* d_offs = nullptr;
  uint64_t n_reads = 0, n_bases = 0;
  std::vector<const uint8_t*> tp;
  for (void* q : d_text) tp.push_back(static_cast<const uint8_t*>(q));
  const bool bam_paired = any_bam && bam.got_paired;
  check(bg_fastq_import(ctx, uint32_t(tp.size()), tp.data(), n_text.data(), pairs.empty() ? 0 : 1, (interleaved || bam_paired) ? 1 : 0, cut_start,
                        cut_end, &d_seq, &d_offs, &n_reads, &n_bases),
        "import");
  for (void* q : d_text) bg_dev_free(ctx, q);
  if (!n_reads) die("\nNo reads were imported, exiting.");
  double t_import = now_s() - t0;
  logf("Total reads imported: %llu (%llu bases, %.2f GB of FASTQ; %.3f s, of which %.3f s moving the files to the GPU)",
       (unsigned long long)n_reads, (unsigned long long)n_bases, text_bytes / 1e9, t_import, t_upload);

  if (prefault.joinable()) prefault.join();
  logf("Running kmerization, read correction, seqset and readmap build on the GPU");
  double t1 = now_s();
  bg_seqset* ss = nullptr;
  bg_readmap* rm = nullptr;
  bg_build_stats st;
  const int is_paired = (!pairs.empty() || interleaved || bam_paired) ? 1 : 0;
  check(bg_create_sample(ctx, d_seq, d_offs, n_reads, &p, is_paired, &ss, &rm, &st), "bg_create_sample");
  double t_build = now_s() - t1;
  bg_dev_free(ctx, d_seq);
  bg_dev_free(ctx, d_offs);
  double pct = double(st.n_corrected_reads) / double(n_reads);
  char msg[256];
  if (pct < min_reads) {  // biograph_create.cpp:748-757
    snprintf(msg, sizeof msg, "Fewer than %2.0f%% of reads (set by --min-reads) were kept after correction (%llu / %llu remain). Cannot continue.",
             min_reads * 100.0, (unsigned long long)st.n_corrected_reads, (unsigned long long)n_reads);
    die(msg);
  }
  if (pct < warn_reads)
    logf("Warning: Fewer than %2.0f%% of reads (set by --warn-reads) survived correction (%llu / %llu remain)",
         warn_reads * 100.0, (unsigned long long)st.n_corrected_reads, (unsigned long long)n_reads);

  mkdirs(out);
  mkdirs(out + "/qc");
  mkdirs(out + "/metadata");
  mkdirs(out + "/coverage");
  double t2 = now_s();
  std::vector<const char*> av(argv, argv + argc);
  // the readmap (download, pack, container, SHA-1 of the file: its name) beside the seqset, over a context and stream of its own
  char sha[41];
  std::string rm_tmp = out + "/coverage/tmp.readmap";
  double t_rm_written = 0;
  std::thread rm_writer([&] {
    bg_ctx* ctx_rm = nullptr;
    check(bg_ctx_create_own_stream(0, &ctx_rm), "bg_ctx_create_own_stream");
    check(bg_readmap_write(ctx_rm, rm, rm_tmp.c_str(), argc, av.data(), sha), "bg_readmap_write");
    t_rm_written = now_s() - t2;
    bg_ctx_destroy(ctx_rm);
  });
  check(bg_seqset_write(ctx, ss, (out + "/seqset").c_str(), argc, av.data()), "bg_seqset_write");
  const double t_ss_written = now_s() - t2;
  rm_writer.join();
  if (rename(rm_tmp.c_str(), (out + "/coverage/" + sha + ".readmap").c_str()) != 0) die("Could not rename " + rm_tmp);
  double t_write = now_s() - t2, t_total = now_s() - t0;
  logf("Written: seqset %.3f s, readmap + digest beside it %.3f s", t_ss_written, t_rm_written);

  std::string id = a.get("id");
  if (id.empty()) id = stem_of(out);
  char uuid[64] = "";
  bg_seqset_uuid(ss, uuid, sizeof uuid);
  char nums[1024];
  snprintf(nums, sizeof nums,
           "\"imported_reads\":%llu,\"imported_bases\":%llu,\"corrected_reads\":%llu,\"corrected_bases\":%llu,"
           "\"avg_bases_per_read\":%.6f,\"corrected_pct\":%.6f,\"kept_kmers\":%llu,\"seqset_entries\":%llu,"
           "\"timings\":[{\"import\":%.3f},{\"gpu_build\":%.3f},{\"write\":%.3f},{\"total\":%.3f}]",
           (unsigned long long)n_reads, (unsigned long long)n_bases, (unsigned long long)st.n_corrected_reads,
           (unsigned long long)st.n_corrected_bases,
           st.n_corrected_reads ? double(st.n_corrected_bases) / double(st.n_corrected_reads) : 0.0, pct,
           (unsigned long long)st.n_kept_kmers, (unsigned long long)st.n_entries, t_import, t_build, t_write, t_total);
  std::string js = "{\"command\":\"create\",\"version\":\"biograph_amd-r2\",\"accession_id\":" + json_str(id) + ",\"uuid\":" +
                   json_str(uuid) + "," + nums + "}";
  write_file(out + "/qc/create_stats.json", js);
  if (a.has("stats")) write_file(a.get("stats"), js);
  std::string meta = "{\"accession_id\":" + json_str(id) + ",\"biograph_id\":" + json_str(uuid) + ",\"command_history\":[" +
                     json_argv(argc, argv) + "],\"samples\":{" + json_str(id) + ":" + json_str(sha) +
                     "},\"version\":\"biograph_amd-r2\"}";
  write_file(out + "/metadata/bg_info.json", meta);
  logf("%llu / %llu reads survived read correction; %llu seqset entries; GPU build %.3f s; %.3f s in all",
       (unsigned long long)st.n_corrected_reads, (unsigned long long)n_reads, (unsigned long long)st.n_entries, t_build, t_total);
  write_file(out + "/qc/create_log.txt", g_log);
  bg_readmap_free(rm);
  bg_seqset_free(ss);
  bg_ctx_destroy(ctx);
  return 0;
}

// bgbinary reference --in X.fasta[.gz] --refdir DIR [--min-n-run 50] [-f]: `biograph reference`
// (modules/main/make_ref.cpp:88-167).  source.fasta = the input (inflated), reference.fasta / karyotype.json /
// reference.ref on the host (bg_reference_build), reference.bwt on the GPU (bg_bwt_build).  The bwa index files the
// reference's wrapper adds with an external `bwa index` are not written.
static int cmd_reference(int argc, char** argv) {
  Args a = parse(argc, argv, 2, {"reads", "refdir", "min-n-run", "force", "no-bwt"});  // --in is parsed as --reads
  if (!a.has("reads") || !a.has("refdir")) die("reference: --in <fasta> and --refdir <ref dir> are required");
  const std::string in = a.get("reads"), dir = a.get("refdir");
  if (exists(dir)) {
    if (!a.has("force")) die("Refusing to overwrite '" + dir + "'. Use --force to override.");
    if (!exists(dir + "/source.fasta")) die(dir + " is not a BioGraph reference. Remove it manually or specify a different location.");
    for (const char* f : {"source.fasta", "reference.fasta", "karyotype.json", "reference.ref", "reference.bwt"}) {
      chmod((dir + "/" + f).c_str(), 0644);
      unlink((dir + "/" + f).c_str());
    }
  }
  mkdirs(dir);
  printf("%s\n", in.size() > 3 && in.substr(in.size() - 3) == ".gz" ? "Unzipping source fasta" : "Preparing source fasta");
  {
    HostText t;
    load_text(in, &t);
    std::string text;
    if (t.fd >= 0) {
      text.resize(t.n);
      size_t got = 0;
      while (got < t.n) {
        const ssize_t r = pread(t.fd, &text[got], t.n - got, off_t(got));
        if (r <= 0) die(in + ": read error");
        got += size_t(r);
      }
    } else {
      text.swap(t.inflated);
    }
    free_text(&t);
    write_file(dir + "/source.fasta", text);
    chmod((dir + "/source.fasta").c_str(), 0444);
  }
  printf("Building reference\n");
  const uint64_t min_n_run = strtoull(a.get("min-n-run", "50").c_str(), nullptr, 10);
  check(bg_reference_build((dir + "/source.fasta").c_str(), dir.c_str(), min_n_run), "bg_reference_build");
  if (!a.has("no-bwt")) {
    bg_reference* rf = nullptr;
    check(bg_reference_open(dir.c_str(), &rf), "bg_reference_open");
    std::string fa(bg_reference_size(rf), '\0');
    check(bg_reference_bases(rf, 0, fa.size(), &fa[0]), "bg_reference_bases");
    std::vector<uint8_t> codes(fa.size(), 0);
    for (size_t i = 0; i < fa.size(); ++i) codes[i] = uint8_t(fa[i] == 'A' ? 0 : fa[i] == 'C' ? 1 : fa[i] == 'G' ? 2 : 3);
    std::vector<uint64_t> es, ez;
    for (uint32_t i = 0; i < bg_reference_supercontig_count(rf); ++i) {
      uint64_t len = 0, flat = 0;
      check(bg_reference_supercontig(rf, i, nullptr, nullptr, &len, &flat), "bg_reference_supercontig");
      es.push_back(flat);
      ez.push_back(len);
    }
    bg_reference_free(rf);
    bg_ctx* ctx = nullptr;
    check(bg_ctx_create(0, nullptr, &ctx), "bg_ctx_create");
    void* d_ref = nullptr;
    check(bg_dev_alloc(ctx, codes.size() ? codes.size() : 1, &d_ref), "bg_dev_alloc");
    check(bg_dev_upload(ctx, d_ref, codes.data(), codes.size()), "upload");
    double t0 = now_s();
    bg_bwt_stats st;
    check(bg_bwt_build(ctx, static_cast<const uint8_t*>(d_ref), uint32_t(es.size()), es.data(), ez.data(), 64,
                       (dir + "/reference.bwt").c_str(), &st), "bg_bwt_build");
    fprintf(stderr, "BWT: %llu entries, %llu century marks, %u doubling rounds, %.3f s\n", (unsigned long long)st.n_entries,
            (unsigned long long)st.n_century, st.doubling_rounds, now_s() - t0);
    bg_dev_free(ctx, d_ref);
    bg_ctx_destroy(ctx);
  }
  printf("Results saved to %s\n", dir.c_str());
  return 0;
}

static int cmd_refmap(int argc, char** argv) {
  Args a = parse(argc, argv, 2, {"reads", "ref", "out"});
  if (!a.has("in") && !a.has("reads")) die("refmap: --in S.bg is required");
  if (!a.has("ref") || !a.has("out")) die("refmap: --ref and --out are required");
  std::string bgdir = a.has("reads") ? a.get("reads") : a.get("in");
  // reference directory (karyotype.json + reference.ref | reference.fasta), .ref or FASTA; extents = its
  // supercontigs = maximal N-free stretches (bio_base/reference.cpp:27-40), walked over the flat genome
  bg_reference* rf = nullptr;
  check(bg_reference_open(a.get("ref").c_str(), &rf), "bg_reference_open");
  std::string fa(bg_reference_size(rf), '\0');
  check(bg_reference_bases(rf, 0, fa.size(), &fa[0]), "bg_reference_bases");
  std::vector<uint8_t> codes(fa.size(), 0);
  for (size_t i = 0; i < fa.size(); ++i) codes[i] = uint8_t(fa[i] == 'A' ? 0 : fa[i] == 'C' ? 1 : fa[i] == 'G' ? 2 : 3);
  std::vector<uint64_t> es, ez;
  for (uint32_t i = 0; i < bg_reference_supercontig_count(rf); ++i) {
    uint64_t len = 0, flat = 0;
    check(bg_reference_supercontig(rf, i, nullptr, nullptr, &len, &flat), "bg_reference_supercontig");
    es.push_back(flat);
    ez.push_back(len);
  }
  bg_reference_free(rf);
  bg_ctx* ctx = nullptr;
  check(bg_ctx_create(0, nullptr, &ctx), "bg_ctx_create");
  bg_seqset* ss = nullptr;
  check(bg_seqset_open(ctx, (bgdir + "/seqset").c_str(), &ss), "bg_seqset_open");
  uint64_t n = bg_seqset_size(ss);
  void *d_ref = nullptr, *d_out = nullptr;
  check(bg_dev_alloc(ctx, codes.size(), &d_ref), "bg_dev_alloc");
  check(bg_dev_alloc(ctx, n, &d_out), "bg_dev_alloc");
  check(bg_dev_upload(ctx, d_ref, codes.data(), codes.size()), "upload");
  double t0 = now_s();
  check(bg_ref_map_build(ctx, ss, static_cast<const uint8_t*>(d_ref), es.size(), es.data(), ez.data(), 0,
                         static_cast<uint8_t*>(d_out)), "bg_ref_map_build");
  double dt = now_s() - t0;
  std::string out(n, '\0');
  check(bg_dev_download(ctx, &out[0], d_out, n), "download");
  write_file(a.get("out"), out);
  uint64_t marked = 0;
  for (unsigned char v : out) marked += (v & 63) != 0;
  fprintf(stderr, "%llu nodes marked by walking reference (%.2f%%) in %.3f s\n", (unsigned long long)marked,
          n ? marked * 100.0 / n : 0.0, dt);
  bg_seqset_free(ss);
  bg_dev_free(ctx, d_ref);
  bg_dev_free(ctx, d_out);
  bg_ctx_destroy(ctx);
  return 0;
}

// Minimal reader of metadata/bg_info.json: accession id and the sample -> readmap sha map.
static std::string read_text(const std::string& p) {
  FILE* f = fopen(p.c_str(), "rb");
  if (!f) die("Could not open " + p);
  std::string out;
  char buf[65536];
  size_t got;
  while ((got = fread(buf, 1, sizeof buf, f)) > 0) out.append(buf, got);
  fclose(f);
  return out;
}
// a JSON string starting at the opening quote js[p]: its unescaped value; *end = index past the closing quote
static std::string json_take_string(const std::string& js, size_t p, size_t* end) {
  std::string v;
  size_t i = p + 1;
  for (; i < js.size() && js[i] != '"'; ++i) {
    if (js[i] == '\\' && i + 1 < js.size()) {
      ++i;
      v.push_back(js[i] == 'n' ? '\n' : js[i] == 't' ? '\t' : js[i]);
    } else {
      v.push_back(js[i]);
    }
  }
  *end = i + 1;
  return v;
}
static std::string json_string(const std::string& js, const std::string& key) {
  std::string pat = "\"" + key + "\":\"";
  size_t p = js.find(pat);
  if (p == std::string::npos) return "";
  size_t e;
  return json_take_string(js, p + pat.size() - 1, &e);
}
static std::map<std::string, std::string> json_samples(const std::string& js) {
  std::map<std::string, std::string> out;
  size_t p = js.find("\"samples\":{");
  if (p == std::string::npos) return out;
  p += 11;
  while (p < js.size()) {
    while (p < js.size() && (js[p] == ',' || js[p] == ' ' || js[p] == '\n')) ++p;
    if (p >= js.size() || js[p] != '"') break;  // '}' closes the object
    size_t e;
    std::string k = json_take_string(js, p, &e);
    size_t v0 = js.find('"', e);
    if (v0 == std::string::npos) break;
    std::string v = json_take_string(js, v0, &e);
    out[k] = v;
    p = e;
  }
  return out;
}

// bgbinary merge --in A.bg --in B.bg [...] --out M.bg [--id ID] [-f]   (biograph_merge.cpp:189-400)
// `merge --gpus N` (biograph_merge.cpp:189-290 over the ranks of a communicator): every rank opens every input and sends its
// stripe of each input's entries through bg_merge_sharded; all ranks write the merged seqset in place (bg_sharded_write);
// the stripes of the merge maps travel to rank 0, which migrates and writes the readmaps and the metadata.
static int merge_rank(const Args& a, int argc, char** argv, const std::vector<std::string>& ins, const std::string& out, const std::string& id_arg,
                      int rank, int world, int device, const std::string& backend, const char* comm_id, int id_fd_in, const std::vector<int>& id_fd_out) {
  const bool lead = rank == 0;
  bg_ctx* ctx = nullptr;
  bg_comm* comm = nullptr;
  rank_open(rank, world, device, backend, comm_id, id_fd_in, id_fd_out, &ctx, &comm);
  const double t0 = now_s();
  std::vector<bg_seqset*> ss(ins.size(), nullptr);
  std::vector<const bg_seqset*> css;
  std::vector<void*> maps(ins.size(), nullptr);
  std::vector<uint64_t*> mapp;
  std::vector<uint64_t> n_in(ins.size()), cnt(ins.size());
  for (size_t i = 0; i < ins.size(); ++i) {
    check(bg_seqset_open(ctx, (ins[i] + "/seqset").c_str(), &ss[i]), "bg_seqset_open");
    css.push_back(ss[i]);
    n_in[i] = bg_seqset_size(ss[i]);
    cnt[i] = n_in[i] / uint64_t(world) + (uint64_t(rank) < n_in[i] % uint64_t(world) ? 1 : 0);  // this rank's stripe (bg_merge_sharded)
    check(bg_dev_alloc(ctx, std::max<uint64_t>(cnt[i], 1) * 8, &maps[i]), "bg_dev_alloc");
    mapp.push_back(static_cast<uint64_t*>(maps[i]));
  }
  bg_sharded* sh = nullptr;
  bg_sharded_stats st;
  check(bg_merge_sharded(ctx, comm, uint32_t(ins.size()), css.data(), &sh, &st, mapp.data()), "bg_merge_sharded");
  if (lead) fprintf(stderr, "Merged seqset: %llu entries on %d rank(s) (%.3f s)\n", (unsigned long long)st.build.n_entries, world, now_s() - t0);
  for (size_t i = 0; i < ins.size(); ++i) bg_seqset_free(ss[i]);
  if (lead) {
    mkdirs(out);
    mkdirs(out + "/qc");
    mkdirs(out + "/metadata");
    mkdirs(out + "/coverage");
  }
  check(bg_comm_barrier(comm), "bg_comm_barrier");  // the directory exists before anybody writes into it
  std::vector<const char*> av(argv, argv + argc);
  char uuid[64] = "";
  check(bg_sharded_write(ctx, comm, sh, (out + "/seqset").c_str(), nullptr, argc, av.data(), uuid, nullptr), "bg_sharded_write");
  bg_sharded_free(sh);
  // the maps: stripe r of input i goes to rank 0 (stripes are in rank order: put together they are the input's map)
  std::vector<void*> full(ins.size(), nullptr);
  for (size_t i = 0; i < ins.size(); ++i) {
    std::vector<uint64_t> send(world, 0), recv(world, 0);
    send[0] = cnt[i];
    if (lead)
      for (int r = 0; r < world; ++r) recv[r] = n_in[i] / uint64_t(world) + (uint64_t(r) < n_in[i] % uint64_t(world) ? 1 : 0);
    if (lead) check(bg_dev_alloc(ctx, std::max<uint64_t>(n_in[i], 1) * 8, &full[i]), "bg_dev_alloc");
    void* dummy = nullptr;
    if (!lead) check(bg_dev_alloc(ctx, 8, &dummy), "bg_dev_alloc");
    check(bg_comm_exchange(comm, maps[i], send.data(), recv.data(), 8, lead ? full[i] : dummy), "bg_comm_exchange");
    if (dummy) bg_dev_free(ctx, dummy);
    bg_dev_free(ctx, maps[i]);
  }
  int rc = 0;
  if (lead) {
    std::vector<std::string> infos;
    std::string merged_id;
    for (size_t i = 0; i < ins.size(); ++i) {
      infos.push_back(read_text(ins[i] + "/metadata/bg_info.json"));
      merged_id += (i ? "+" : "") + json_string(infos[i], "accession_id");  // biograph_merge.cpp:334-342
    }
    if (!id_arg.empty()) merged_id = id_arg;
    std::string samples_json;
    size_t n_samples = 0;
    std::map<std::string, bool> seen_samples;
    for (size_t i = 0; i < ins.size(); ++i) {
      for (auto& kv : json_samples(infos[i])) {
        if (seen_samples.count(kv.first)) {  // biograph_merge.cpp:116-134
          fprintf(stderr, "Sample %s of %s is already present in the merge: skipped\n", kv.first.c_str(), ins[i].c_str());
          continue;
        }
        seen_samples[kv.first] = true;
        bg_readmap *rm = nullptr, *mig = nullptr;
        check(bg_readmap_open(ctx, (ins[i] + "/coverage/" + kv.second + ".readmap").c_str(), &rm), "bg_readmap_open");
        check(bg_readmap_migrate_to(ctx, rm, static_cast<const uint64_t*>(full[i]), st.build.n_entries, uuid, &mig), "bg_readmap_migrate_to");
        char sha[41];
        std::string tmp = out + "/coverage/tmp.readmap";
        check(bg_readmap_write(ctx, mig, tmp.c_str(), argc, av.data(), sha), "bg_readmap_write");
        if (rename(tmp.c_str(), (out + "/coverage/" + sha + ".readmap").c_str()) != 0) die("Could not rename " + tmp);
        samples_json += std::string(n_samples ? "," : "") + json_str(kv.first) + ":" + json_str(sha);
        ++n_samples;
        bg_readmap_free(mig);
        bg_readmap_free(rm);
      }
    }
    std::string meta = "{\"accession_id\":" + json_str(merged_id) + ",\"biograph_id\":" + json_str(uuid) + ",\"command_history\":[" +
                       json_argv(argc, argv) + "],\"samples\":{" + samples_json + "},\"version\":\"biograph_amd-r2\"}";
    write_file(out + "/metadata/bg_info.json", meta);
    std::string js = "{\"command\":\"merge\",\"version\":\"biograph_amd-r2\",\"accession_id\":" + json_str(merged_id) +
                     ",\"samples\":" + std::to_string(n_samples) + ",\"seqset_entries\":" + std::to_string(st.build.n_entries) +
                     ",\"gpus\":" + std::to_string(world) + ",\"comm\":" + json_str(backend) + "}";
    write_file(out + "/qc/merge_stats.json", js);
    fprintf(stderr, "\n%s created.\n", out.c_str());
    for (void* f : full) if (f) bg_dev_free(ctx, f);
  }
  check(bg_comm_barrier(comm), "bg_comm_barrier");
  bg_comm_free(comm);
  bg_ctx_destroy(ctx);
  return rc;
}

static int cmd_merge(int argc, char** argv) {
  Args a = parse(argc, argv, 2, {"reads", "out", "id", "force", "gpus", "comm", "devices", "force-sharded"});
  if (!a.has("reads") || !a.has("out")) die("merge: --in (at least once) and --out are required");
  const auto& ins = a.opt["reads"];  // the parser files --in under --reads
  std::string out = a.get("out");
  if (out.size() < 3 || out.substr(out.size() - 3) != ".bg") out += ".bg";
  if (exists(out + "/seqset") && !a.has("force")) die("Refusing to overwrite " + out + ". Use --force to override.");
  const int gpus = atoi(a.get("gpus", "1").c_str());
  if (gpus < 1 || gpus > 127) die("--gpus takes 1..127");
  if (gpus > 1 || a.has("force-sharded")) {
    const std::string id_arg = a.has("id") ? a.get("id") : std::string();
    return launch_ranks(a, gpus, [&](int r, int device, const std::string& backend, const char* comm_id, int in, const std::vector<int>& outs) {
      return merge_rank(a, argc, argv, ins, out, id_arg, r, gpus, device, backend, comm_id, in, outs);
    });
  }
  bg_ctx* ctx = nullptr;
  check(bg_ctx_create(0, nullptr, &ctx), "bg_ctx_create");
  std::vector<bg_seqset*> ss(ins.size(), nullptr);
  std::vector<const bg_seqset*> css;
  std::vector<void*> maps(ins.size(), nullptr);
  std::vector<uint64_t*> mapp;
  std::vector<std::string> infos;
  std::string merged_id;
  for (size_t i = 0; i < ins.size(); ++i) {
    check(bg_seqset_open(ctx, (ins[i] + "/seqset").c_str(), &ss[i]), "bg_seqset_open");
    css.push_back(ss[i]);
    check(bg_dev_alloc(ctx, std::max<uint64_t>(bg_seqset_size(ss[i]), 1) * 8, &maps[i]), "bg_dev_alloc");
    mapp.push_back(static_cast<uint64_t*>(maps[i]));
    infos.push_back(read_text(ins[i] + "/metadata/bg_info.json"));
    std::string acc = json_string(infos[i], "accession_id");
    merged_id += (i ? "+" : "") + acc;  // :334-342
  }
  if (a.has("id")) merged_id = a.get("id");
  double t0 = now_s();
  bg_seqset* merged = nullptr;
  check(bg_seqset_merge(ctx, uint32_t(ins.size()), css.data(), &merged, mapp.data()), "bg_seqset_merge");
  fprintf(stderr, "Merged seqset: %llu entries (%.3f s)\n", (unsigned long long)bg_seqset_size(merged), now_s() - t0);
  mkdirs(out);
  mkdirs(out + "/qc");
  mkdirs(out + "/metadata");
  mkdirs(out + "/coverage");
  std::vector<const char*> av(argv, argv + argc);
  check(bg_seqset_write(ctx, merged, (out + "/seqset").c_str(), argc, av.data()), "bg_seqset_write");
  std::string samples_json;
  size_t n_samples = 0;
  std::map<std::string, bool> seen_samples;
  for (size_t i = 0; i < ins.size(); ++i) {
    for (auto& kv : json_samples(infos[i])) {
      if (seen_samples.count(kv.first)) {  // biograph_merge.cpp:116-134: a sample that is already in is skipped
        fprintf(stderr, "Sample %s of %s is already present in the merge: skipped\n", kv.first.c_str(), ins[i].c_str());
        continue;
      }
      seen_samples[kv.first] = true;
      bg_readmap *rm = nullptr, *mig = nullptr;
      check(bg_readmap_open(ctx, (ins[i] + "/coverage/" + kv.second + ".readmap").c_str(), &rm), "bg_readmap_open");
      check(bg_readmap_migrate(ctx, rm, mapp[i], merged, &mig), "bg_readmap_migrate");
      char sha[41];
      std::string tmp = out + "/coverage/tmp.readmap";
      check(bg_readmap_write(ctx, mig, tmp.c_str(), argc, av.data(), sha), "bg_readmap_write");
      if (rename(tmp.c_str(), (out + "/coverage/" + sha + ".readmap").c_str()) != 0) die("Could not rename " + tmp);
      samples_json += std::string(n_samples ? "," : "") + json_str(kv.first) + ":" + json_str(sha);
      ++n_samples;
      bg_readmap_free(mig);
      bg_readmap_free(rm);
    }
  }
  char uuid[64] = "";
  bg_seqset_uuid(merged, uuid, sizeof uuid);
  std::string meta = "{\"accession_id\":" + json_str(merged_id) + ",\"biograph_id\":" + json_str(uuid) + ",\"command_history\":[" +
                     json_argv(argc, argv) + "],\"samples\":{" + samples_json + "},\"version\":\"biograph_amd-r2\"}";
  write_file(out + "/metadata/bg_info.json", meta);
  std::string js = "{\"command\":\"merge\",\"version\":\"biograph_amd-r2\",\"accession_id\":" + json_str(merged_id) +
                   ",\"samples\":" + std::to_string(n_samples) + ",\"seqset_entries\":" + std::to_string(bg_seqset_size(merged)) + "}";
  write_file(out + "/qc/merge_stats.json", js);
  fprintf(stderr, "\n%s created.\n", out.c_str());
  bg_seqset_free(merged);
  for (size_t i = 0; i < ins.size(); ++i) { bg_seqset_free(ss[i]); bg_dev_free(ctx, maps[i]); }
  bg_ctx_destroy(ctx);
  return 0;
}

// bgbinary discovery --in S.bg --ref DIR|FASTA --out V.vcf [--sample ID] [--min-overlap 0.7|N] [--max-pair-distance 1000]
// The graph-discover pipeline of python/biograph/variants/graph_discover_test.py:41-56 (make_ref_assemblies ->
// graph_trim_ref -> update_rc_seqset_entries -> discover_branch -> ... -> discover_push_to_pair) per scaffold;
// every anchored non-reference assembly becomes one VCF record.  (The reference's `bgbinary discovery` runs
// the production tracer + genotyper on top of the same walks: SURVEY §8f.2, not built.)
// The reference's discovery engine over one sample (biograph_variants.cpp:598-830 with the push tracer): ref_map, one
// tracer session, trace_ref's work list (scaffold_split_size pieces, both directions; trace_ref.cpp:27-105), then per
// scaffold the serial pipeline (pipeline.cpp:40-57,92-101) and one VCF record per surviving variant
// (ploidless_vcf_export.cpp).  PI is the record's ordinal (the reference's process-wide assembly counter is not
// reproducible); the header's static lines are the reference's, the source line names this program.
// A rank of `discovery --gpus N` (trace_ref.cpp:44-63,105-140 over processes instead of threads): the pieces of every scaffold
// go round-robin to the ranks, each traces its pieces (push forward, push reverse, the two pop tracers) on its GPU, the
// assemblies of a scaffold travel to the rank that owns it (scaffold index % world), which runs the serial pipeline, the
// coverage step and the genotype filter there; rank 0 puts the records together in scaffold order.
struct RankEnv {
  int rank = 0, world = 1, device = 0;
  bg_comm* comm = nullptr;
};

struct Blob {
  std::string b;
  template <class T> void put(const T& v) { b.append(reinterpret_cast<const char*>(&v), sizeof(T)); }
  void put_u32s(const uint32_t* p, uint64_t n) { put(n); if (n) b.append(reinterpret_cast<const char*>(p), n * 4); }
};
struct BlobIn {
  const char* p;
  const char* e;
  template <class T> T get() {
    if (size_t(e - p) < sizeof(T)) die("discovery --gpus: a rank's assemblies arrived cut short");
    T v;
    memcpy(&v, p, sizeof(T));
    p += sizeof(T);
    return v;
  }
  const char* bytes(uint64_t n) {
    if (uint64_t(e - p) < n) die("discovery --gpus: a rank's assemblies arrived cut short");
    const char* q = p;
    p += n;
    return q;
  }
};

// every assembly of a work item's result, as bg_trace_result_get shows it.  Assembly ids come from a per-process counter: id *
// world + rank keeps them apart over the ranks (rvg_exclude groups sections by them).
static void put_result(Blob* out, const bg_trace_result* r, int rank, int world) {
  const uint64_t n = r ? bg_trace_result_size(r) : 0;
  out->put(n);
  for (uint64_t i = 0; i < n; ++i) {
    bg_trace_assembly x;
    check(bg_trace_result_get(r, i, &x), "bg_trace_result_get");
    out->put(x.left_offset); out->put(x.right_offset); out->put(x.score);
    out->put(x.left_anchor_len); out->put(x.right_anchor_len);
    out->put(x.min_overlap); out->put(x.trace_steps); out->put(x.left_anchor_ambiguous_bases);
    out->put(x.matches_reference); out->put(x.from_pop);
    out->put(uint64_t(x.assembly_id * uint64_t(world) + uint64_t(rank)));
    out->put(x.seq_len);
    out->b.append(x.seq, x.seq_len);
    out->put_u32s(x.rc_read_ids, x.n_rc_read_ids);
    out->put_u32s(x.left_pair_matches, x.n_left_pair_matches);
    out->put_u32s(x.right_pair_matches, x.n_right_pair_matches);
  }
}
static bg_trace_result* get_result(BlobIn* in) {
  bg_trace_result* r = nullptr;
  check(bg_trace_result_new(&r), "bg_trace_result_new");
  const uint64_t n = in->get<uint64_t>();
  for (uint64_t i = 0; i < n; ++i) {
    bg_trace_assembly x;
    memset(&x, 0, sizeof x);
    x.left_offset = in->get<int64_t>(); x.right_offset = in->get<int64_t>(); x.score = in->get<int64_t>();
    x.left_anchor_len = in->get<int32_t>(); x.right_anchor_len = in->get<int32_t>();
    x.min_overlap = in->get<uint32_t>(); x.trace_steps = in->get<uint32_t>(); x.left_anchor_ambiguous_bases = in->get<uint32_t>();
    x.matches_reference = in->get<int32_t>(); x.from_pop = in->get<int32_t>();
    x.assembly_id = in->get<uint64_t>();
    x.seq_len = in->get<uint64_t>();
    x.seq = in->bytes(x.seq_len);
    std::vector<uint32_t> ids[3];
    for (int k = 0; k < 3; ++k) {
      const uint64_t m = in->get<uint64_t>();
      const char* q = in->bytes(m * 4);
      ids[k].resize(m);
      if (m) memcpy(ids[k].data(), q, m * 4);
    }
    x.rc_read_ids = ids[0].data(); x.n_rc_read_ids = ids[0].size();
    x.left_pair_matches = ids[1].data(); x.n_left_pair_matches = ids[1].size();
    x.right_pair_matches = ids[2].data(); x.n_right_pair_matches = ids[2].size();
    check(bg_trace_result_add(r, &x), "bg_trace_result_add");
  }
  return r;
}

// every rank's bytes on every rank (sizes over the host, the bytes through the communicator's device all-gather)
static std::vector<std::string> gather_blobs(const RankEnv& env, bg_ctx* ctx, const std::string& mine) {
  std::vector<uint64_t> sizes(size_t(env.world), 0);
  const uint64_t n = mine.size();
  check(bg_comm_all_gather_host(env.comm, &n, 8, sizes.data()), "bg_comm_all_gather_host");
  uint64_t total = 0;
  for (uint64_t v : sizes) total += v;
  void *d_send = nullptr, *d_recv = nullptr;
  check(bg_dev_alloc(ctx, n ? n : 1, &d_send), "bg_dev_alloc");
  check(bg_dev_alloc(ctx, total ? total : 1, &d_recv), "bg_dev_alloc");
  if (n) check(bg_dev_upload(ctx, d_send, mine.data(), n), "upload");
  check(bg_comm_all_gather(env.comm, d_send, sizes.data(), d_recv), "bg_comm_all_gather");
  std::string all(total, '\0');
  if (total) check(bg_dev_download(ctx, &all[0], d_recv, total), "download");
  bg_dev_free(ctx, d_send);
  bg_dev_free(ctx, d_recv);
  std::vector<std::string> parts;
  uint64_t at = 0;
  for (uint64_t v : sizes) { parts.push_back(all.substr(at, v)); at += v; }
  return parts;
}

// how many of the n bytes at p are A, C, G or T before the first one that is not (a chromosome is a run of 60 million: 32 bytes
// a step where the processor has AVX2; the scalar loop took a tenth of a second of the command's serial part)
static inline bool is_acgt(char c) { return c == 'A' || c == 'C' || c == 'G' || c == 'T'; }
#if defined(__x86_64__)
__attribute__((target("avx2"))) static size_t acgt_run_avx2(const char* p, size_t n) {
  const __m256i A = _mm256_set1_epi8('A'), C = _mm256_set1_epi8('C'), G = _mm256_set1_epi8('G'), T = _mm256_set1_epi8('T');
  size_t i = 0;
  for (; i + 32 <= n; i += 32) {
    const __m256i v = _mm256_loadu_si256(reinterpret_cast<const __m256i*>(p + i));
    const __m256i m = _mm256_or_si256(_mm256_or_si256(_mm256_cmpeq_epi8(v, A), _mm256_cmpeq_epi8(v, C)),
                                      _mm256_or_si256(_mm256_cmpeq_epi8(v, G), _mm256_cmpeq_epi8(v, T)));
    const uint32_t ok = uint32_t(_mm256_movemask_epi8(m));
    if (ok != 0xffffffffu) return i + size_t(__builtin_ctz(~ok));
  }
  while (i < n && is_acgt(p[i])) ++i;
  return i;
}
#endif
static size_t acgt_run(const char* p, size_t n) {
#if defined(__x86_64__)
  static const bool avx2 = __builtin_cpu_supports("avx2");
  if (avx2) return acgt_run_avx2(p, n);
#endif
  size_t i = 0;
  while (i < n && is_acgt(p[i])) ++i;
  return i;
}

static int discovery_push(Args& a, const std::vector<std::pair<std::string, std::string>>& recs,
                          const std::map<std::string, std::vector<std::pair<int64_t, int64_t>>>& bed, bg_ctx* ctx, bg_seqset* ss,
                          bg_readmap* rm, uint32_t min_overlap, int64_t max_pair_distance, const std::string& sample,
                          const std::vector<uint8_t>& flat_codes, const std::vector<uint64_t>& flat_start,
                          const std::vector<uint64_t>& flat_size, const RankEnv& env) {
  const double t0 = now_s();
  const uint64_t n = bg_seqset_size(ss);
  const bool ranks = env.world > 1 || env.comm != nullptr;  // (--force-sharded: one rank through the same path)
  void *d_ref = nullptr, *d_map = nullptr;
  check(bg_dev_alloc(ctx, flat_codes.size() ? flat_codes.size() : 1, &d_ref), "bg_dev_alloc");
  check(bg_dev_alloc(ctx, n ? n : 1, &d_map), "bg_dev_alloc");
  // --ref-map FILE (biograph_variants.cpp:171,643-666): the reference map kept between runs -- built into FILE.new and
  // renamed when FILE does not exist, then opened; "no validity checking is done to make sure this matches the current
  // reference and seqset" beyond its size (ref_map.cpp: CHECK_EQ(m_ref_map->size(), m_seqset->size()))
  const std::string ref_map_file = a.get("ref-map");
  const bool ref_map_on_file = !ref_map_file.empty() && exists(ref_map_file);
  if (!ref_map_on_file) {
    check(bg_dev_upload(ctx, d_ref, flat_codes.data(), flat_codes.size()), "upload");
    check(bg_ref_map_build(ctx, ss, static_cast<const uint8_t*>(d_ref), flat_start.size(), flat_start.data(), flat_size.data(), 0,
                           static_cast<uint8_t*>(d_map)), "bg_ref_map_build");
    if (!ref_map_file.empty() && env.rank == 0) {  // (ranks: the first one writes it, every rank keeps what it built)
      std::vector<uint8_t> h(n);
      check(bg_dev_download(ctx, h.data(), d_map, n), "download");
      const std::string fresh = ref_map_file + ".new";
      unlink(fresh.c_str());
      fprintf(stderr, "Generating refmap in %s\n", fresh.c_str());
      const char* av[] = {"bgbinary", "discovery"};
      check(bg_ref_map_write(fresh.c_str(), h.data(), n, 2, av), "bg_ref_map_write");
      if (rename(fresh.c_str(), ref_map_file.c_str()) != 0) die("Could not rename " + fresh);
    }
  }
  if (ref_map_on_file || (!ref_map_file.empty() && !ranks)) {
    fprintf(stderr, "Opening refmap %s\n", ref_map_file.c_str());
    std::vector<uint8_t> h(n ? n : 1);
    check(bg_ref_map_read(ref_map_file.c_str(), h.data(), n), "bg_ref_map_read");
    check(bg_dev_upload(ctx, d_map, h.data(), n), "upload");
  }
  bg_dev_free(ctx, d_ref);
  const double t_refmap = now_s() - t0;
  bg_trace_options opt;
  bg_trace_options_default(&opt);
  opt.min_overlap = min_overlap;
  if (a.has("min-pop-overlap")) {  // biograph_variants.cpp:198-200,687
    const long long v = atoll(a.get("min-pop-overlap").c_str());
    if (v < 1) die("discovery: --min-pop-overlap must be at least 1");
    opt.min_pop_overlap = uint64_t(v);
  }
  // --assemblies-out FILE (biograph_variants.cpp:166,540-565,781-788): every assembly that enters the pipeline, behind its
  // sorter (report_discovered_assemblies, pipeline.cpp:22-54), as CSV
  std::string asm_csv;
  const bool want_csv = a.has("assemblies-out") && !a.get("assemblies-out").empty();
  if (want_csv)
    asm_csv = "scaffold_name,left_offset,right_offset,left_anchor_len,right_anchor_len,aid,score,min_overlap,ref_seq,seq,generated_by\n";
  // --aligned-assemblies-out FILE (biograph_variants.cpp:168,566-597,735-744): the assemblies behind the aligner (align.cpp:89-90)
  // with their aligned variants as refrange:varseq:refseq
  std::string aligned_csv;
  const bool want_aligned = a.has("aligned-assemblies-out") && !a.get("aligned-assemblies-out").empty();
  if (want_aligned)
    aligned_csv = "scaffold_name,left_offset,right_offset,left_anchor_len,right_anchor_len,aid,score,min_overlap,"
                  "variants(refrange:varseq:refseq),seq,generated_by\n";
  // --half-aligned-out FILE (biograph_variants.cpp:175,695-710): what the anchor_dropper could not place on the right
  std::string half_csv;
  const bool want_half = a.has("half-aligned-out") && !a.get("half-aligned-out").empty();
  if (want_half) half_csv = "scaffold_name,left_anchor,right_anchor,sequence,assembly_id\n";
  if (ranks && (want_csv || want_aligned || want_half))
    die("discovery --gpus: --assemblies-out, --aligned-assemblies-out and --half-aligned-out are written by the single-process command only");
  opt.max_pair_distance = uint64_t(max_pair_distance);
  if (a.has("scaffold-split-size")) {  // assemble_options::scaffold_split_size (the size of trace_ref's pieces)
    const long long sp = atoll(a.get("scaffold-split-size").c_str());
    if (sp < 1000) die("discovery: --scaffold-split-size must be at least 1000");
    opt.scaffold_split_size = uint64_t(sp);
  }
  bg_coverage_options cov;
  bg_coverage_options_default(&cov);
  // biograph_variants.cpp:686 always writes the ML feature columns (LASCORE ... NUMASM); --ml-features false leaves them out
  {
    const std::string ml = a.get("ml-features", "true");
    const int on = !(ml == "false" || ml == "0" || ml == "off" || ml == "no");
    opt.output_ml_features = on;
    cov.output_ml_features = on;
  }
  // --simple-gt (biograph_variants.cpp:201,690; false there: every covered variant section is written, unfiltered) and
  // --rvg-exclude (:194,689; true there: rvg_exclude.cpp behind the genotype step, pipeline.cpp:102-104)
  auto flag = [&](const char* name, bool dflt) {
    if (!a.has(name)) return dflt;
    const std::string v = a.get(name);
    if (v == "true" || v == "1" || v == "on" || v == "yes") return true;
    if (v == "false" || v == "0" || v == "off" || v == "no") return false;
    die(std::string("discovery: --") + name + " takes true or false");
    return dflt;
  };
  // --enable-pop-tracer (biograph_variants.cpp:188,685; true there): the pop tracer re-traces the push tracer's anchor drops
  const bool pop_tracer = flag("enable-pop-tracer", true);
  cov.simple_genotype_filter = flag("simple-gt", false) ? 1 : 0;
  cov.rvg_exclude = flag("rvg-exclude", true) ? 1 : 0;
  // trace_ref runs its work items on the process's thread pool (trace_ref.cpp:105-140); here: --threads sessions
  // (default 16, or the host's cores when fewer: since the traces that never branch run on the device -- all but six in a thousand -- and
  // the search states come out of a thread-caching allocator, tracing the chr20-shaped sample takes 8.6 s on 8 sessions,
  // 5.7 s on 16, 5.9-6.4 s on 24, 7.1 s on 32, 8.7 s on 48: profiles/r04_discovery_sessions_cpu_time.txt; round 3: 12.9 s on
  // 16, 11.3 s on 24), each with its own context (stream + allocator) and scratch, the read-only host tables shared, side
  // by side on the one GPU
  int n_threads = a.has("threads") ? atoi(a.get("threads").c_str())
                                   : int(std::min(16u, std::max(1u, std::thread::hardware_concurrency())));
  if (n_threads < 1) n_threads = 1;
  if (n_threads > 64) n_threads = 64;
  const double t_open0 = now_s();
  // (the coverage session's tables beside the tracer's: both are downloads of readmap columns and host loops over them)
  // (on a context of its own: the steps behind the tracers of one scaffold run beside the tracing of the next)
  bg_coverage* cov_session = nullptr;
  std::thread cov_open([&] {
    bg_ctx* ctx_cov = nullptr;
    check(bg_ctx_create_own_stream(env.device, &ctx_cov), "bg_ctx_create_own_stream");
    check(bg_coverage_open(ctx_cov, ss, rm, &cov_session), "bg_coverage_open");
  });
  std::vector<bg_ctx*> wctx(size_t(n_threads), nullptr);
  std::vector<bg_tracer*> tracers(size_t(n_threads), nullptr);
  wctx[0] = ctx;
  const double t_first0 = now_s();
  check(bg_tracer_open(wctx[0], ss, rm, static_cast<const uint8_t*>(d_map), &opt, &tracers[0]), "bg_tracer_open");
  const double t_first = now_s() - t_first0;
  {  // the further sessions (a context, a stream, page-locked buffers each) side by side
    std::vector<std::thread> openers;
    for (int t = 1; t < n_threads; ++t)
      openers.emplace_back([&, t] {
        check(bg_ctx_create_own_stream(env.device, &wctx[size_t(t)]), "bg_ctx_create_own_stream");
        check(bg_tracer_clone(tracers[0], wctx[size_t(t)], &tracers[size_t(t)]), "bg_tracer_clone");
      });
    for (auto& th : openers) th.join();
  }
  const double t_clones = now_s() - t_first0 - t_first;
  cov_open.join();
  const double t_open = now_s() - t_open0;
  if (getenv("BG_TRACER_TIMING")) fprintf(stderr, "discovery: first session %.3f s, the other %d %.3f s, coverage session through %.3f s after the start of all\n", t_first, n_threads - 1, t_clones, t_open);
  time_t now = time(nullptr);
  char date[16];
  strftime(date, sizeof date, "%Y%m%d", localtime(&now));
  std::string vcf = std::string("##fileformat=VCFv4.1\n##fileDate=") + date + "\n##source=\"biograph_amd bgbinary discovery (push tracer engine)\"\n" +
                    "##reference=" + a.get("ref") + "\n"
                    "##INFO=<ID=NS,Number=1,Type=Integer,Description=\"Number of Samples\">\n"
                    "##INFO=<ID=SVTYPE,Number=1,Type=String,Description=\"Structural Variant Type\">\n"
                    "##INFO=<ID=SVLEN,Number=1,Type=Integer,Description=\"Difference in length between REF and ALT alleles\">\n"
                    "##INFO=<ID=END,Number=1,Type=Integer,Description=\"End position of the variant described in this record\">\n"
                    "##FORMAT=<ID=GT,Number=1,Type=String,Description=\"Genotype\">\n"
                    "##FORMAT=<ID=DP,Number=1,Type=Integer,Description=\"Sample Depth\">\n"
                    "##FORMAT=<ID=AD,Number=.,Type=Integer,Description=\"Allelic depths for the ref and alt alleles in the order listed\">\n"
                    "##FORMAT=<ID=PDP,Number=1,Type=Integer,Description=\"Sample Pair Depth\">\n"
                    "##FORMAT=<ID=PAD,Number=.,Type=Integer,Description=\"Allelic pair depths for the ref and alt alleles in the order listed\">\n"
                    "##FORMAT=<ID=OV,Number=1,Type=Integer,Description=\"Minimum read overlap in assembly\">\n"
                    "##FORMAT=<ID=PG,Number=1,Type=String,Description=\"Phased genotype\">\n"
                    "##FORMAT=<ID=PI,Number=1,Type=Integer,Description=\"Phase group\">\n"
                    "##FORMAT=<ID=GQ,Number=1,Type=Integer,Description=\"Genotype quality\">\n"
                    + (cov.output_ml_features ? std::string(
                    "##FORMAT=<ID=LASCORE,Number=1,Type=Integer,Description=\"Score of longest assembly\">\n"
                    "##FORMAT=<ID=LAREFSPAN,Number=1,Type=Integer,Description=\"Ref span length of longest assembly\">\n"
                    "##FORMAT=<ID=LARANCH,Number=1,Type=Integer,Description=\"Right anchor length of longest assembly\">\n"
                    "##FORMAT=<ID=LALANCH,Number=1,Type=Integer,Description=\"Leftanchor length of longest assembly\">\n"
                    "##FORMAT=<ID=LAREFGC,Number=1,Type=Float,Description=\"Portion of G/C bases in ref span of longest assembly\">\n"
                    "##FORMAT=<ID=LAALTGC,Number=1,Type=Float,Description=\"Portion of G/C bases in sequence of longest assembly\">\n"
                    "##FORMAT=<ID=LAALTSEQLEN,Number=1,Type=Integer,Description=\"Sequence length of longest assembly\">\n"
                    "##FORMAT=<ID=NUMASM,Number=1,Type=Integer,Description=\"Number of assemblies that independently produced this variant\">\n")
                                                : std::string()) +
                    "##ALT=<ID=INS,Description=\"Insertion\">\n##ALT=<ID=DEL,Description=\"Deletion\">\n";
  for (auto& rec : recs) vcf += "##contig=<ID=" + rec.first + ",length=" + std::to_string(rec.second.size()) + ">\n";
  vcf += "#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\t" + sample + "\n";
  uint64_t n_var = 0, n_traced = 0, n_sections = 0, pieces = 0, n_popped = 0;
  double t_trace = 0, t_pipe = 0, t_cov = 0, t_vcf = 0;
  double rss_trace = 0, rss_pipe = 0, rss_cov = 0;
  std::thread finisher;  // the scaffold before the one being traced, behind its tracers
  // the coverage session takes one scaffold's sections at a time: a scaffold whose sections are fed behind its tracers
  // (bg_coverage_feed) waits for the scaffold before it to be through with the session
  std::mutex cov_turn_mu;
  std::condition_variable cov_turn_cv;
  size_t cov_turn = 0, cov_tickets = 0;
  const double rss_sessions = rss_gb();
  struct ScaffoldData {
    std::vector<int64_t> eo;
    std::vector<std::string_view> es;
    std::vector<const char*> ep;
    std::vector<uint64_t> el;
    std::vector<bg_trace_result*> items;
    const std::string* name;
    const std::string* seq;
  };
  // the VCF record lines of a scaffold's kept sections, numbered from `before` + 1 on (ploidless_vcf_export: PI); how many
  auto format_records = [&](const bg_trace_result* kept, const ScaffoldData& d, uint64_t before, std::string* out) {
    std::vector<char> buf(1 << 16);
    uint64_t made = 0;
    for (uint64_t i = 0; i < bg_trace_result_size(kept); ++i) {
      uint64_t len = 0;
      check(bg_vcf_record(kept, i, d.name->c_str(), d.es.size(), d.eo.data(), d.ep.data(), d.el.data(), int64_t(d.seq->size()), &cov, before + made + 1,
                          buf.data(), buf.size(), &len), "bg_vcf_record");
      if (len + 1 > buf.size()) {
        buf.resize(len + 1);
        check(bg_vcf_record(kept, i, d.name->c_str(), d.es.size(), d.eo.data(), d.ep.data(), d.el.data(), int64_t(d.seq->size()), &cov, before + made + 1,
                            buf.data(), buf.size(), &len), "bg_vcf_record");
      }
      if (!len) continue;
      out->append(buf.data(), len);
      ++made;
    }
    return made;
  };
  struct Held { size_t scaffold; bg_trace_result* kept; std::shared_ptr<ScaffoldData> sd; uint64_t records; };
  std::vector<Held> held;  // (ranks) the scaffolds this rank owns, until their record numbers are known
  for (auto& rec : recs) {
    const std::string& seq = rec.second;
    std::vector<int64_t> eo;
    std::vector<std::string_view> es;  // extents = maximal ACGT stretches (supercontigs), where they lie in the scaffold
    for (size_t i = 0; i < seq.size();) {
      const size_t run = acgt_run(seq.data() + i, seq.size() - i);
      if (run) { eo.push_back(int64_t(i)); es.push_back(std::string_view(seq).substr(i, run)); }
      i += run;
      while (i < seq.size() && !is_acgt(seq[i])) ++i;
    }
    if (es.empty()) continue;
    std::vector<const char*> ep;
    std::vector<uint64_t> el;
    for (auto& x : es) { ep.push_back(x.data()); el.push_back(x.size()); }
    std::vector<std::pair<int64_t, int64_t>> ranges;  // trace_ref::add_scaffold_range
    if (a.has("bed")) {
      auto it = bed.find(rec.first);
      if (it == bed.end()) continue;
      for (auto& iv : it->second) ranges.emplace_back(iv.first, std::min<int64_t>(iv.second, int64_t(seq.size())));
    } else {
      ranges.emplace_back(0, int64_t(seq.size()));
    }
    struct Work { int64_t start, limit; int rev; };
    std::vector<Work> work;
    std::vector<size_t> my;  // the work items this rank traces: the pieces go round-robin to the ranks, both directions of a piece together
    for (auto& rg : ranges)
      for (int64_t start = rg.first; start < rg.second; start += int64_t(opt.scaffold_split_size)) {
        const int64_t limit = std::min<int64_t>(rg.second, start + int64_t(opt.scaffold_split_size));
        const bool mine = pieces % uint64_t(env.world) == uint64_t(env.rank);
        ++pieces;
        for (int rev = 0; rev < 2; ++rev) {
          if (mine) my.push_back(work.size());
          work.push_back(Work{start, limit, rev});
        }
      }
    std::vector<bg_trace_result*> items(work.size(), nullptr);
    std::vector<bg_trace_result*> pops(pop_tracer ? work.size() : 0, nullptr);  // (piece, direction) like `work`
    double t1 = now_s();
    // One rank: a piece's pop tracers run as soon as its two push directions are through (not after every piece's), and the
    // serial pipeline is fed the finished pieces in order while the sessions trace the rest (bg_pipeline_feed: what lies before a
    // mark no later piece reaches goes through the steps; a piece that breaks a mark sends everything through
    // bg_assemblies_process as before).  BG_DISCOVERY_STREAM=0: first every push, then every pop, then the pipeline.
    static const bool stream_off = [] { const char* e = getenv("BG_DISCOVERY_STREAM"); return e && e[0] == '0'; }();
    // BG_DISCOVERY_STREAM=coverage: what leaves the pipeline goes on to the coverage session as well, in a few large stretches
    // (measured on the chr20-shaped sample and left off: the coverage step's sixty threads and their device rounds beside the
    // sessions slow the tracing from 1.55 to 2.0 s for the 0.2 s of coverage they hide: 3.3 s for the command against 3.0-3.2)
    static const bool cov_stream_off = [] { const char* e = getenv("BG_DISCOVERY_STREAM"); return !(e && e[0] == 'c'); }();
    bg_trace_result* streamed_sections = nullptr;
    double t_stream_tail = 0;
    bool cov_fed = false;           // the coverage session holds this scaffold's sections so far
    uint64_t n_sections_fed = 0;
    const size_t cov_ticket = (ranks && !(int(size_t(&rec - &recs[0]) % size_t(env.world)) == env.rank)) ? size_t(-1) : cov_tickets++;
    if (!ranks && !stream_off && !want_csv && !want_aligned && !work.empty()) {
      const size_t n_pieces = work.size() / 2;
      // how far before its piece an assembly may start: the reverse direction reads read_ahead_distance behind the piece, a pop
      // tracer's fronts lie within an assembly's length and a pair's distance of the drops (generous: a broken mark only costs time)
      int64_t reach = int64_t(opt.read_ahead_distance) + 4 * int64_t(opt.max_pair_distance) + 8 * int64_t(bg_seqset_max_len(ss)) + 4000;
      if (const char* e = getenv("BG_DISCOVERY_REACH")) reach = atoll(e);  // (tests: marks that later pieces break)
      std::vector<int64_t> mark(n_pieces, INT64_MAX);  // after piece p: no later piece starts before mark[p]
      for (size_t p = n_pieces; p-- > 1;) mark[p - 1] = std::min(mark[p], work[2 * p].start - reach);
      std::mutex mu;
      std::condition_variable cv;
      std::deque<size_t> pop_ready;          // work items whose pop tracer may run
      std::vector<int> push_done(n_pieces, 0), pop_done(n_pieces, 0);
      std::vector<char> piece_done(n_pieces, 0);
      std::vector<std::unique_ptr<std::once_flag>> waited;
      for (size_t p = 0; p < n_pieces; ++p) waited.emplace_back(new std::once_flag);
      size_t next_push = 0, pops_taken = 0;
      const size_t n_pops = pop_tracer ? work.size() : 0;
      std::string err;
      bool failed = false;
      auto fail_with = [&](const std::string& what) {
        std::lock_guard<std::mutex> l(mu);
        if (!failed) { failed = true; err = what; }
        cv.notify_all();
      };
      // (the anchor_dropper and the pair_counter of a push item run on its session's helper thread: whoever needs the result waits)
      auto wait_pushes = [&](size_t p) {
        std::call_once(*waited[p], [&] {
          for (size_t i = 2 * p; i < 2 * p + 2; ++i)
            if (bg_trace_result_wait(items[i]) != BG_OK) fail_with(std::string("bg_tracer_trace: ") + bg_last_error());
        });
      };
      auto worker = [&](int t) {
        static const bool sync_only = [] { const char* e = getenv("BG_TRACER_ASYNC"); return e && e[0] == '0'; }();
        for (;;) {
          size_t i = 0;
          bool is_pop = false;
          {
            std::unique_lock<std::mutex> l(mu);
            for (;;) {
              if (failed) return;
              if (!pop_ready.empty()) { i = pop_ready.front(); pop_ready.pop_front(); ++pops_taken; is_pop = true; break; }
              if (next_push < work.size()) { i = next_push++; break; }
              if (pops_taken >= n_pops) return;
              cv.wait(l);
            }
          }
          const size_t p = i / 2;
          if (!is_pop) {
            if ((sync_only ? bg_tracer_trace : bg_tracer_trace_async)(tracers[size_t(t)], es.size(), eo.data(), ep.data(), el.data(), int64_t(seq.size()),
                                                                      work[i].start, work[i].limit, work[i].rev, 1, &items[i]) != BG_OK) {
              fail_with(std::string("bg_tracer_trace: ") + bg_last_error());
              return;
            }
            std::lock_guard<std::mutex> l(mu);
            if (++push_done[p] == 2) {
              if (pop_tracer) { pop_ready.push_back(2 * p); pop_ready.push_back(2 * p + 1); }
              else piece_done[p] = 1;
              cv.notify_all();
            }
          } else {
            wait_pushes(p);
            {
              std::lock_guard<std::mutex> l(mu);
              if (failed) return;
            }
            const bg_trace_result* push[2] = {items[2 * p], items[2 * p + 1]};
            if (bg_pop_trace(tracers[size_t(t)], es.size(), eo.data(), ep.data(), el.data(), int64_t(seq.size()), work[i].start, work[i].limit,
                             work[i].rev, push, 2, &pops[i]) != BG_OK) {
              fail_with(std::string("bg_pop_trace: ") + bg_last_error());
              return;
            }
            std::lock_guard<std::mutex> l(mu);
            if (++pop_done[p] == 2) { piece_done[p] = 1; cv.notify_all(); }
          }
        }
      };
      // the pipeline behind the pieces, in piece order
      bg_pipeline* pl = nullptr;
      check(bg_pipeline_open(es.size(), eo.data(), ep.data(), el.data(), int64_t(seq.size()), &opt, BG_STEP_PIPELINE, &pl), "bg_pipeline_open");
      // (the normalizer lets go of what lies this far before the variant coming in, instead of keeping a whole extent -- a
      // chromosome -- to the end: the coverage step gets its sections behind the tracers.  A variant that moves further left
      // through a repeat breaks that like a mark.  BG_DISCOVERY_EARLY=bases, negative: off)
      int64_t early = 20000;
      if (const char* e = getenv("BG_DISCOVERY_EARLY")) early = atoll(e);
      if (early >= 0 && !cov_stream_off) check(bg_pipeline_release_early(pl, early), "bg_pipeline_release_early");
      bool marks_held = true;
      std::string feed_err;
      std::thread feeder([&] {
        int64_t promised = INT64_MIN;  // (the last piece promises nothing new: what it leaves goes through at the finish)
        for (size_t p = 0; p < n_pieces; ++p) {
          {
            std::unique_lock<std::mutex> l(mu);
            cv.wait(l, [&] { return failed || piece_done[p]; });
            if (failed) return;
          }
          if (!pop_tracer) wait_pushes(p);
          const bg_trace_result* in[4] = {items[2 * p], items[2 * p + 1], pop_tracer ? pops[2 * p] : nullptr, pop_tracer ? pops[2 * p + 1] : nullptr};
          int behind = 0;
          if (mark[p] != INT64_MAX) promised = std::max(promised, mark[p]);
          if (bg_pipeline_feed(pl, in, pop_tracer ? 4 : 2, promised, &behind) != BG_OK) { feed_err = bg_last_error(); marks_held = false; return; }
          if (behind) { marks_held = false; return; }
          // (the coverage step runs its islands on dozens of threads in rounds on the device: a piece's few hundred islands at a
          // time would cost every piece the rounds of a whole scaffold -- what has left the pipeline goes on in a few large stretches)
          static const uint64_t cov_batch = [] { const char* e = getenv("BG_DISCOVERY_COV_BATCH"); return e ? uint64_t(atoll(e)) : uint64_t(20000); }();
          if (cov_stream_off || bg_pipeline_ready(pl) < cov_batch) continue;
          bg_trace_result* stretch = nullptr;
          if (bg_pipeline_take(pl, &stretch) != BG_OK) { feed_err = bg_last_error(); marks_held = false; return; }
          if (bg_trace_result_size(stretch)) {
            if (!cov_fed) {  // (the scaffold before this one may still be with the session)
              std::unique_lock<std::mutex> l(cov_turn_mu);
              cov_turn_cv.wait(l, [&] { return cov_turn == cov_ticket; });
              bg_coverage_discard(cov_session);
              cov_fed = true;
            }
            n_sections_fed += bg_trace_result_size(stretch);
            if (bg_coverage_feed(cov_session, stretch, es.size(), eo.data(), ep.data(), el.data(), int64_t(seq.size()), &cov, 1, 0, nullptr) != BG_OK) {
              feed_err = bg_last_error();
              marks_held = false;
              bg_trace_result_free(stretch);
              return;
            }
          }
          bg_trace_result_free(stretch);
        }
      });
      const int nt = int(std::min<size_t>(size_t(n_threads), work.size()));
      std::vector<std::thread> pool;
      for (int t = 1; t < nt; ++t) pool.emplace_back(worker, t);
      worker(0);
      for (auto& th : pool) th.join();
      const double t_traced = now_s();
      { std::lock_guard<std::mutex> l(mu); cv.notify_all(); }
      feeder.join();
      if (failed) die(err);
      if (!pop_tracer)
        for (size_t p = 0; p < n_pieces; ++p) wait_pushes(p);
      if (failed) die(err);
      if (!feed_err.empty()) die("bg_pipeline_feed: " + feed_err);
      if (marks_held) {
        check(bg_pipeline_finish(pl, &streamed_sections), "bg_pipeline_finish");
        if (!streamed_sections) marks_held = false;  // (the tail broke the normalizer's early release)
      }
      if (!marks_held)
        fprintf(stderr, "discovery: a piece's assemblies lay further before it than %lld bases, or a variant moved further left than %lld: the "
                        "scaffold's pipeline runs over all of them at once\n", (long long)reach, (long long)early);
      bg_pipeline_close(pl);
      t_stream_tail = now_s() - t_traced;
      if (pop_tracer) {
        std::vector<bg_trace_result*> all;
        for (size_t i = 0; i + 1 < work.size(); i += 2) {
          all.push_back(items[i]); all.push_back(items[i + 1]); all.push_back(pops[i]); all.push_back(pops[i + 1]);
        }
        for (auto* r : pops) if (r) n_popped += bg_trace_result_size(r);
        items.swap(all);
      }
    } else
    {
      std::atomic<size_t> next(0);
      std::mutex err_mu;
      std::string err;
      auto run = [&](int t) {
        for (;;) {
          const size_t at = next.fetch_add(1);
          if (at >= my.size()) return;
          const size_t i = my[at];
          // (the anchor_dropper and the pair_counter of this item run on the session's helper thread while the session traces its next)
          static const bool sync_only = [] { const char* e = getenv("BG_TRACER_ASYNC"); return e && e[0] == '0'; }();
          if ((sync_only ? bg_tracer_trace : bg_tracer_trace_async)(tracers[size_t(t)], es.size(), eo.data(), ep.data(), el.data(), int64_t(seq.size()),
                                                                    work[i].start, work[i].limit, work[i].rev, 1, &items[i]) != BG_OK) {
            std::lock_guard<std::mutex> l(err_mu);
            if (err.empty()) err = bg_last_error();
            next.store(my.size());
            return;
          }
        }
      };
      const int nt = int(std::min<size_t>(size_t(n_threads), std::max<size_t>(1, my.size())));
      std::vector<std::thread> pool;
      for (int t = 1; t < nt; ++t) pool.emplace_back(run, t);
      run(0);
      for (auto& th : pool) th.join();
      if (!err.empty()) die("bg_tracer_trace: " + err);
      for (auto* r : items)
        if (r) check(bg_trace_result_wait(r), "bg_tracer_trace");
      // the pop tracers of every piece (trace_ref.cpp:186-207): forward and reverse, over the anchor drops both push
      // directions of the piece reported
      if (pop_tracer) {
        next.store(0);
        auto run_pop = [&](int t) {
          for (;;) {
            const size_t at = next.fetch_add(1);
            if (at >= my.size()) return;
            const size_t i = my[at];
            const bg_trace_result* push[2] = {items[i & ~size_t(1)], items[i | 1]};
            if (bg_pop_trace(tracers[size_t(t)], es.size(), eo.data(), ep.data(), el.data(), int64_t(seq.size()), work[i].start,
                             work[i].limit, work[i].rev, push, 2, &pops[i]) != BG_OK) {
              std::lock_guard<std::mutex> l(err_mu);
              if (err.empty()) err = bg_last_error();
              next.store(my.size());
              return;
            }
          }
        };
        std::vector<std::thread> pool2;
        for (int t = 1; t < nt; ++t) pool2.emplace_back(run_pop, t);
        run_pop(0);
        for (auto& th : pool2) th.join();
        if (!err.empty()) die("bg_pop_trace: " + err);
        // the order the assemblies of a work item reach the scaffold's pipeline in: push forward, push reverse, pop forward,
        // pop reverse
        std::vector<bg_trace_result*> all;
        for (size_t i = 0; i + 1 < work.size(); i += 2) {
          all.push_back(items[i]); all.push_back(items[i + 1]); all.push_back(pops[i]); all.push_back(pops[i + 1]);
        }
        for (auto* r : pops) if (r) n_popped += bg_trace_result_size(r);
        items.swap(all);
      }
    }
    for (auto* r : items) if (r) n_traced += bg_trace_result_size(r);
    const size_t scaffold_k = size_t(&rec - &recs[0]);
    const bool owner = int(scaffold_k % size_t(env.world)) == env.rank;
    if (ranks) {
      // this rank's pieces of the scaffold to its owner: per piece its index and its results in pipeline order
      const size_t per = pop_tracer ? 4 : 2;
      Blob mine;
      uint64_t n_mine = 0;
      for (size_t pi = 0; pi * per < items.size(); ++pi) n_mine += items[pi * per] ? 1 : 0;
      mine.put(n_mine);
      for (size_t pi = 0; pi * per < items.size(); ++pi) {
        if (!items[pi * per]) continue;
        mine.put(uint64_t(pi));
        for (size_t j = 0; j < per; ++j) put_result(&mine, items[pi * per + j], env.rank, env.world);
      }
      for (auto*& r : items) { if (r) bg_trace_result_free(r); r = nullptr; }
      const std::vector<std::string> parts = gather_blobs(env, ctx, mine.b);
      if (owner) {
        for (const std::string& part : parts) {
          BlobIn in{part.data(), part.data() + part.size()};
          const uint64_t np = in.get<uint64_t>();
          for (uint64_t q = 0; q < np; ++q) {
            const uint64_t pi = in.get<uint64_t>();
            if (pi * per >= items.size() || items[pi * per]) die("discovery --gpus: a piece arrived twice or from nowhere");
            for (size_t j = 0; j < per; ++j) items[pi * per + j] = get_result(&in);
          }
        }
        for (auto* r : items) if (!r) die("discovery --gpus: a piece of the scaffold did not arrive");
      }
    }
    if (want_half)  // (the reference writes them as its threads report them; here in work-item order)
      for (auto* r : items)
        for (uint64_t i = 0; i < bg_trace_result_half_aligned_count(r); ++i) {
          int64_t off = 0;
          int right = 0;
          const char* hs = nullptr;
          uint64_t hn = 0, id = 0;
          check(bg_trace_result_half_aligned(r, i, &off, &right, &hs, &hn, &id), "bg_trace_result_half_aligned");
          half_csv += rec.first + "," + (right ? "," + std::to_string(off) : std::to_string(off) + ",") + ",";
          half_csv.append(hs, hn);
          half_csv += "," + std::to_string(id) + "\n";
        }
    t_trace += now_s() - t1 - t_stream_tail;  // (what the fed pipeline still had to do when the tracers were through is the pipeline's)
    rss_trace = std::max(rss_trace, rss_gb());
    // The steps behind the tracers -- pipeline, coverage, VCF records -- of this scaffold on a thread of their own while the
    // sessions trace the next scaffold (BG_DISCOVERY_OVERLAP=0: first to the end); one scaffold at a time, in order: the
    // records are numbered across scaffolds.
    if (ranks && !owner) continue;  // (its pieces are with the scaffold's owner)
    if (finisher.joinable()) finisher.join();
    std::shared_ptr<ScaffoldData> sd(new ScaffoldData);
    sd->eo.swap(eo);
    sd->es.swap(es);
    for (auto& x : sd->es) { sd->ep.push_back(x.data()); sd->el.push_back(x.size()); }
    sd->items.swap(items);
    sd->name = &rec.first;
    sd->seq = &rec.second;
    bg_trace_result* const streamed = streamed_sections;
    const double stream_tail = t_stream_tail;
    const bool cov_streamed = cov_fed && streamed;  // (marks that broke after a stretch went on: bg_coverage_run starts the scaffold over)
    const uint64_t sections_fed = cov_streamed ? n_sections_fed : 0;
    finisher = std::thread([&, sd, scaffold_k, streamed, stream_tail, cov_streamed, sections_fed, cov_ticket] {
    auto& eo = sd->eo;
    auto& es = sd->es;
    auto& ep = sd->ep;
    auto& el = sd->el;
    auto& items = sd->items;
    const std::string& seq = *sd->seq;
    struct { const std::string& first; } rec{*sd->name};
    double t1;
    t1 = now_s();
    bg_trace_result *sections = nullptr, *kept = nullptr;
    if (want_csv) {
      bg_trace_result* sorted = nullptr;
      check(bg_assemblies_process(items.data(), uint32_t(items.size()), es.size(), eo.data(), ep.data(), el.data(), int64_t(seq.size()),
                                  &opt, BG_STEP_SORT, &sorted), "bg_assemblies_process");
      for (uint64_t i = 0; i < bg_trace_result_size(sorted); ++i) {
        bg_trace_assembly x;
        check(bg_trace_result_get(sorted, i, &x), "bg_trace_result_get");
        char head[256];
        snprintf(head, sizeof head, "%s,%lld,%lld,%d,%d,%llu,%lld,%u,", rec.first.c_str(), (long long)x.left_offset, (long long)x.right_offset,
                 x.left_anchor_len, x.right_anchor_len, (unsigned long long)x.assembly_id, (long long)x.score, x.min_overlap);
        asm_csv += head;
        if (x.left_offset >= 0 && x.right_offset >= x.left_offset && uint64_t(x.right_offset) <= seq.size())
          asm_csv.append(seq, size_t(x.left_offset), size_t(x.right_offset - x.left_offset));  // scaffold::subscaffold_str: N in the gaps
        asm_csv += ",";
        asm_csv.append(x.seq, x.seq_len);
        asm_csv += x.from_pop ? ",POP\n" : ",PUSH\n";
      }
      bg_trace_result_free(sorted);
    }
    if (want_aligned) {
      bg_trace_result* al = nullptr;
      check(bg_assemblies_process(items.data(), uint32_t(items.size()), es.size(), eo.data(), ep.data(), el.data(), int64_t(seq.size()),
                                  &opt, BG_STEP_SORT | BG_STEP_TRIM_REF | BG_STEP_DEDUP | BG_STEP_ALIGN, &al), "bg_assemblies_process");
      for (uint64_t i = 0; i < bg_trace_result_size(al); ++i) {
        bg_trace_assembly x;
        check(bg_trace_result_get(al, i, &x), "bg_trace_result_get");
        char head[256];
        snprintf(head, sizeof head, "%s,%lld,%lld,%d,%d,%llu,%lld,%u,", rec.first.c_str(), (long long)x.left_offset, (long long)x.right_offset,
                 x.left_anchor_len, x.right_anchor_len, (unsigned long long)x.assembly_id, (long long)x.score, x.min_overlap);
        aligned_csv += head;
        const uint64_t nv = bg_trace_result_var_count(al, i);
        for (uint64_t j = 0; j < nv; ++j) {
          int64_t vl = 0, vr = 0;
          const char* vs = nullptr;
          uint64_t vn = 0;
          check(bg_trace_result_var(al, i, j, &vl, &vr, &vs, &vn), "bg_trace_result_var");
          if (j) aligned_csv += ";";
          aligned_csv += std::to_string(vl) + "-" + std::to_string(vr) + ":";
          aligned_csv.append(vs, vn);
          aligned_csv += ":";
          if (vl >= 0 && vr >= vl && uint64_t(vr) <= seq.size()) aligned_csv.append(seq, size_t(vl), size_t(vr - vl));
        }
        aligned_csv += ",";
        aligned_csv.append(x.seq, x.seq_len);
        aligned_csv += x.from_pop ? ",POP\n" : ",PUSH\n";
      }
      bg_trace_result_free(al);
    }
    if (streamed) sections = streamed;  // (fed piece by piece behind the tracers; its tail is counted below)
    else
      check(bg_assemblies_process(items.data(), uint32_t(items.size()), es.size(), eo.data(), ep.data(), el.data(), int64_t(seq.size()),
                                  &opt, BG_STEP_PIPELINE, &sections), "bg_assemblies_process");
    for (auto* r : items) bg_trace_result_free(r);
    n_sections += bg_trace_result_size(sections) + sections_fed;
    t_pipe += now_s() - t1 + stream_tail;
    rss_pipe = std::max(rss_pipe, rss_gb());
    t1 = now_s();
    {  // (this scaffold's turn with the session, if it has not taken it behind its tracers already)
      std::unique_lock<std::mutex> l(cov_turn_mu);
      cov_turn_cv.wait(l, [&] { return cov_turn == cov_ticket; });
    }
    if (cov_streamed)  // the stretches before went on behind the tracers: what the pipeline's finish left, and the scaffold is through
      check(bg_coverage_feed(cov_session, sections, es.size(), eo.data(), ep.data(), el.data(), int64_t(seq.size()), &cov, 1, 1, &kept), "bg_coverage_feed");
    else
      check(bg_coverage_run(cov_session, sections, es.size(), eo.data(), ep.data(), el.data(), int64_t(seq.size()), &cov, 1, &kept),
            "bg_coverage_run");
    {
      std::lock_guard<std::mutex> l(cov_turn_mu);
      ++cov_turn;
      cov_turn_cv.notify_all();
    }
    bg_trace_result_free(sections);
    t_cov += now_s() - t1;
    rss_cov = std::max(rss_cov, rss_gb());
    t1 = now_s();
    if (ranks) {
      // the records are numbered across scaffolds: how many this one has now, the lines when every owner has said so
      std::string scratch;
      held.push_back(Held{scaffold_k, kept, sd, format_records(kept, *sd, 0, &scratch)});
    } else {
      n_var += format_records(kept, *sd, n_var, &vcf);
      bg_trace_result_free(kept);
    }
    t_vcf += now_s() - t1;
    });
    static const bool overlap = [] { const char* e = getenv("BG_DISCOVERY_OVERLAP"); return !(e && e[0] == '0'); }();
    if (!overlap) finisher.join();
  }
  if (finisher.joinable()) finisher.join();
  if (ranks) {
    // how many records every scaffold has (its owner knows) -> where each scaffold's numbers start -> the lines -> rank 0
    std::vector<uint64_t> cnt(recs.size(), 0), every(recs.size() * size_t(env.world), 0), before(recs.size() + 1, 0);
    for (const Held& h : held) cnt[h.scaffold] = h.records;
    if (!recs.empty()) check(bg_comm_all_gather_host(env.comm, cnt.data(), recs.size() * 8, every.data()), "bg_comm_all_gather_host");
    for (size_t k = 0; k < recs.size(); ++k) {
      uint64_t c = 0;
      for (int r = 0; r < env.world; ++r) c += every[size_t(r) * recs.size() + k];
      before[k + 1] = before[k] + c;
    }
    Blob lines;
    lines.put(uint64_t(held.size()));
    for (const Held& h : held) {
      std::string text;
      if (format_records(h.kept, *h.sd, before[h.scaffold], &text) != h.records) die("discovery --gpus: a scaffold's records changed between two passes");
      bg_trace_result_free(h.kept);
      lines.put(uint64_t(h.scaffold));
      lines.put(uint64_t(text.size()));
      lines.b += text;
    }
    const std::vector<std::string> parts = gather_blobs(env, ctx, lines.b);
    uint64_t sums[3] = {n_traced, n_popped, n_sections}, all_sums[3 * 64];
    if (env.world > 64) die("discovery --gpus: at most 64 ranks");
    check(bg_comm_all_gather_host(env.comm, sums, sizeof sums, all_sums), "bg_comm_all_gather_host");
    check(bg_comm_barrier(env.comm), "bg_comm_barrier");  // (nobody leaves while another rank still reads what it sent)
    if (env.rank != 0) return 0;
    n_traced = n_popped = n_sections = 0;
    for (int r = 0; r < env.world; ++r) { n_traced += all_sums[3 * r]; n_popped += all_sums[3 * r + 1]; n_sections += all_sums[3 * r + 2]; }
    std::vector<std::string> text(recs.size());
    for (const std::string& part : parts) {
      BlobIn in{part.data(), part.data() + part.size()};
      const uint64_t nh = in.get<uint64_t>();
      for (uint64_t q = 0; q < nh; ++q) {
        const uint64_t k = in.get<uint64_t>(), len = in.get<uint64_t>();
        if (k >= recs.size()) die("discovery --gpus: records of an unknown scaffold");
        text[k].assign(in.bytes(len), len);
      }
    }
    for (const std::string& t : text) vcf += t;
    n_var = before[recs.size()];
  }
  // (the sessions, their contexts and the reference map are not closed one by one: the process ends when the outputs are written,
  // and handing sixteen sessions' pinned buffers, arenas and streams back first took a tenth of a second)
  const double t_write0 = now_s();
  write_file(a.get("out"), vcf);
  if (want_csv) write_file(a.get("assemblies-out"), asm_csv);
  if (want_aligned) write_file(a.get("aligned-assemblies-out"), aligned_csv);
  if (want_half) write_file(a.get("half-aligned-out"), half_csv);
  fprintf(stderr, "%llu variants over %zu scaffold(s) in %llu piece(s) x 2 directions on %d tracer session(s), min overlap %u: %.2f s (refmap %.2f, sessions %.2f, tracing %.2f "
                  "[%llu assemblies], pipeline %.2f [%llu sections], coverage + genotype %.2f, VCF records %.2f); pop tracer %s [%llu of the assemblies]\n",
          (unsigned long long)n_var, recs.size(), (unsigned long long)pieces, n_threads, min_overlap, now_s() - t0, t_refmap, t_open, t_trace,
          (unsigned long long)n_traced, t_pipe, (unsigned long long)n_sections, t_cov, t_vcf, pop_tracer ? "on" : "off", (unsigned long long)n_popped);
  fprintf(stderr, "discovery: outputs written in %.2f s; %.2f s since the process started; resident with the sessions open %.1f GB, after "
                  "tracing %.1f, after the pipeline %.1f, after coverage %.1f, now %.1f\n", now_s() - t_write0,
          since_process_start_s(), rss_sessions, rss_trace, rss_pipe, rss_cov, rss_gb());
  return 0;
}

static int discovery_main(Args& a, const std::string& engine, int rank, int world, int device, const std::string& backend, const char* comm_id,
                          int id_fd_in, const std::vector<int>& id_fd_out);

static int cmd_discovery(int argc, char** argv) {
  // (these settings are this command's alone: `create` holds multi-gigabyte FASTQ text on the heap and wants it back)
  mallopt(M_TRIM_THRESHOLD, 1 << 30);
  mallopt(M_MMAP_THRESHOLD, 1 << 30);
  mallopt(M_TOP_PAD, 256 << 20);
  Args a = parse(argc, argv, 2, {"reads", "ref", "out", "sample", "min-overlap", "max-pair-distance", "bed", "engine", "max-ploids",
                                 "enable-pop-tracer", "threads", "ml-features", "scaffold-split-size", "simple-gt", "rvg-exclude",
                                 "force", "ref-map", "min-pop-overlap", "assemblies-out", "aligned-assemblies-out", "half-aligned-out",
                                 "gpus", "comm", "devices", "force-sharded"});
  if (!a.has("reads") || !a.has("ref") || !a.has("out")) die("discovery: --in S.bg, --ref and --out are required");
  // biograph_variants.cpp:216-232: an output that exists is only replaced with -f / --force
  for (const char* o : {"out", "assemblies-out", "aligned-assemblies-out", "half-aligned-out"})
    if (a.has(o) && !a.get(o).empty() && a.get(o) != "-" && exists(a.get(o))) {
      if (!a.has("force")) die("Refusing to overwrite '" + a.get(o) + "'. Use -f to override.");
      unlink(a.get(o).c_str());
    }
  // --engine push (default): the reference's discovery engine -- push tracer, anchor_dropper, pair_counter, the
  // serial pipeline, calc_coverage, simple_genotype_filter, ploidless VCF records (biograph_variants.cpp:598-830);
  // --engine graph: the graph_discover pipeline (branch + push_to_pair), positions and alleles only
  const std::string engine = a.get("engine", "push");
  if (engine != "push" && engine != "graph") die("discovery: --engine takes push or graph");
  // --max-ploids (biograph_variants.cpp:162) feeds genotyper / ploid_limiter, neither of which is a step of the standard
  // pipeline (pipeline.cpp:92-105): accepted, without effect there either.
  if (a.has("max-ploids") && atoi(a.get("max-ploids").c_str()) < 1) die("discovery: --max-ploids must be at least 1");
  // --gpus N [--comm rccl|shm] [--devices a,b,...]: the work items of trace_ref (trace_ref.cpp:44-63) over N processes, one per GPU,
  // each with the whole biograph on its device; --gpus 1 --force-sharded: the same path on one rank
  const int gpus = a.has("gpus") ? atoi(a.get("gpus").c_str()) : 1;
  if (gpus < 1) die("discovery: --gpus must be at least 1");
  if (gpus > 1 || a.has("force-sharded")) {
    if (engine != "push") die("discovery --gpus: the push engine only");
    return launch_ranks(a, gpus, [&](int r, int device, const std::string& backend, const char* comm_id, int in, const std::vector<int>& outs) {
      return discovery_main(a, engine, r, gpus, device, backend, comm_id, in, outs);
    });
  }
  return discovery_main(a, engine, 0, 1, 0, "", nullptr, -1, {});
}

// All the command's threads on the CPUs of ONE memory node (the one this thread runs on, or BG_PIN_NODE=n; =off: as the scheduler
// likes).  The discovery engine is a few dozen threads chasing pointers through a gigabyte of tables that the opening thread
// touched first: on the two-socket host of the GPU boxes a thread on the other socket pays the far memory for every miss --
// the chr20-shaped sample traces in 1.86 s instead of 2.04 when the process stays on either socket (tools/discovery_numa.sh;
// 32 of its cores are as good as all 64).  Memory follows by first touch.  A host with one node: nothing happens.
static void pin_to_one_memory_node() {
  const char* e = getenv("BG_PIN_NODE");
  if (e && !strcmp(e, "off")) return;
  int node = -1;
  auto cpus_of = [](int n, cpu_set_t* out) -> bool {
    char path[96];
    snprintf(path, sizeof path, "/sys/devices/system/node/node%d/cpulist", n);
    FILE* f = fopen(path, "r");
    if (!f) return false;
    char buf[4096];
    const bool got = fgets(buf, sizeof buf, f) != nullptr;
    fclose(f);
    if (!got) return false;
    CPU_ZERO(out);
    for (char* p = buf; *p && *p != '\n';) {
      char* q = nullptr;
      const long lo = strtol(p, &q, 10);
      if (q == p) break;
      long hi = lo;
      p = q;
      if (*p == '-') { hi = strtol(p + 1, &q, 10); p = q; }
      for (long c = lo; c <= hi && c < CPU_SETSIZE; ++c) CPU_SET(int(c), out);
      if (*p == ',') ++p;
    }
    return true;
  };
  cpu_set_t have, of_node, want;
  if (sched_getaffinity(0, sizeof have, &have) != 0) return;
  if (e && *e >= '0' && *e <= '9') {
    node = atoi(e);
  } else {
    const int cpu = sched_getcpu();
    if (cpu < 0) return;
    for (int n = 0; n < 64 && node < 0; ++n)
      if (cpus_of(n, &of_node) && CPU_ISSET(cpu, &of_node)) node = n;
  }
  if (node < 0 || !cpus_of(node, &of_node)) return;
  CPU_AND(&want, &have, &of_node);
  if (CPU_COUNT(&want) < 8 || CPU_COUNT(&want) == CPU_COUNT(&have)) return;  // (too few to run the sessions on, or one node anyway)
  (void)sched_setaffinity(0, sizeof want, &want);
}

static int discovery_main(Args& a, const std::string& engine, int rank, int world, int device, const std::string& backend, const char* comm_id,
                          int id_fd_in, const std::vector<int>& id_fd_out) {
  if (!comm_id || getenv("BG_PIN_NODE")) pin_to_one_memory_node();  // (before the first thread is made: they inherit the mask; the ranks of --gpus N are left where they are unless told)
  cpuprof::start();
  const std::string bgdir = a.get("reads");
  RankEnv env;
  env.rank = rank;
  env.world = world;
  // scaffolds (chromosomes) and their N-free supercontigs from the reference directory / .ref / FASTA; a
  // scaffold is rebuilt with N between its supercontigs so that VCF positions are scaffold coordinates
  const double t_begin = now_s();
  // the reference (a host-only read and unpack of its own) on a thread beside the seqset and the readmap going to the device
  std::vector<std::pair<std::string, std::string>> recs;
  std::vector<uint8_t> flat_codes;
  std::vector<uint64_t> flat_start, flat_size;
  double t_ref = 0;
  std::thread ref_thread([&] {
    bg_reference* rf = nullptr;
    check(bg_reference_open(a.get("ref").c_str(), &rf), "bg_reference_open");
    for (uint32_t c = 0; c < bg_reference_scaffold_count(rf); ++c)
      recs.emplace_back(bg_reference_scaffold_name(rf, c), std::string(bg_reference_scaffold_len(rf, c), 'N'));
    for (uint32_t i = 0; i < bg_reference_supercontig_count(rf); ++i) {
      uint32_t sc = 0;
      uint64_t off = 0, len = 0, flat = 0;
      check(bg_reference_supercontig(rf, i, &sc, &off, &len, &flat), "bg_reference_supercontig");
      if (sc >= recs.size() || off > recs[sc].second.size() || len > recs[sc].second.size() - off)
        die("reference: a supercontig lies outside its scaffold (corrupt reference?)");
      check(bg_reference_bases(rf, flat, len, &recs[sc].second[off]), "bg_reference_bases");
    }
    // the flat genome and its supercontigs, for the ref_map of the push engine (ref_map::build, ref_map.cpp:72-185)
    if (engine == "push") {
      std::string fa(bg_reference_size(rf), '\0');
      check(bg_reference_bases(rf, 0, fa.size(), &fa[0]), "bg_reference_bases");
      flat_codes.resize(fa.size());
      for (size_t i = 0; i < fa.size(); ++i) flat_codes[i] = uint8_t((uint8_t(fa[i]) >> 1 ^ uint8_t(fa[i]) >> 2) & 3);  // A 0, C 1, G 2, T 3 (the bases are upper-case ACGT)
      for (uint32_t i = 0; i < bg_reference_supercontig_count(rf); ++i) {
        uint64_t len = 0, flat = 0;
        check(bg_reference_supercontig(rf, i, nullptr, nullptr, &len, &flat), "bg_reference_supercontig");
        flat_start.push_back(flat);
        flat_size.push_back(len);
      }
    }
    bg_reference_free(rf);
    t_ref = now_s();
  });
  bg_ctx* ctx = nullptr;
  if (comm_id) {  // a rank: its device, its context and the communicator
    const int ndev = bg_device_count();
    if (ndev < 1) die("--gpus: no HIP device");
    env.device = device >= 0 ? device : (backend == "shm" ? rank % ndev : rank);
    rank_open(rank, world, env.device, backend, comm_id, id_fd_in, id_fd_out, &ctx, &env.comm);
  } else {
    check(bg_ctx_create(0, nullptr, &ctx), "bg_ctx_create");
  }
  const double t_ctx = now_s();
  auto samples = json_samples(read_text(bgdir + "/metadata/bg_info.json"));
  std::string sample = a.get("sample");
  if (samples.empty()) die("No samples found in BioGraph metadata.");                           // biograph_dir.cpp:104-107
  if (sample.empty()) {
    if (samples.size() > 1) die("Multiple samples are present. You must specify a readmap or accession ID.");  // :108-111
    sample = samples.begin()->first;
  }
  if (!samples.count(sample)) die("Sample " + sample + " not found in " + bgdir);
  // the readmap beside the seqset: its unpacking is host work, its upload goes over a stream of its own (a second context, which
  // lives as long as the process)
  bg_readmap* rm = nullptr;
  double t_rm_done = 0;
  std::thread rm_thread([&] {
    bg_ctx* ctx_rm = nullptr;
    check(bg_ctx_create_own_stream(env.device, &ctx_rm), "bg_ctx_create_own_stream");
    check(bg_readmap_open(ctx_rm, (bgdir + "/coverage/" + samples[sample] + ".readmap").c_str(), &rm), "bg_readmap_open");
    t_rm_done = now_s();
  });
  bg_seqset* ss = nullptr;
  check(bg_seqset_open(ctx, (bgdir + "/seqset").c_str(), &ss), "bg_seqset_open");
  const double t_ss = now_s();
  rm_thread.join();
  double mo = atof(a.get("min-overlap", "0.7").c_str());
  uint32_t min_overlap = mo < 1.0 ? uint32_t(mo * bg_readmap_max_read_len(rm)) : uint32_t(mo);  // biograph_variants.cpp:674
  int64_t max_pair_distance = atoll(a.get("max-pair-distance", "1000").c_str());
  // --bed FILE: only these regions are walked (chrom, start, end; biograph_variants.cpp:144-211)
  std::map<std::string, std::vector<std::pair<int64_t, int64_t>>> bed;
  if (a.has("bed")) {
    std::string txt = read_text(a.get("bed"));
    size_t p0 = 0;
    while (p0 < txt.size()) {
      size_t e = txt.find('\n', p0);
      if (e == std::string::npos) e = txt.size();
      std::string line = txt.substr(p0, e - p0);
      p0 = e + 1;
      if (line.empty() || line[0] == '#' || line.compare(0, 5, "track") == 0 || line.compare(0, 7, "browser") == 0) continue;
      char chrom[256];
      long long b0 = 0, b1 = 0;
      if (sscanf(line.c_str(), "%255s %lld %lld", chrom, &b0, &b1) != 3 || b1 <= b0 || b0 < 0) die("--bed: malformed line: " + line);
      bed[chrom].emplace_back(b0, b1);
    }
  }
  const double t_rm = now_s();
  ref_thread.join();
  double t0 = now_s();
  fprintf(stderr, "discovery: reference, seqset and readmap open in %.2f s (reference %.2f on a thread of its own beside: device context %.2f, "
                  "seqset %.2f, readmap %.2f beside it on a stream of its own; %.2f s from the start of the process to here, resident %.1f GB)\n",
          t0 - t_begin, t_ref - t_begin, t_ctx - t_begin, t_ss - t_ctx, t_rm_done - t_ctx, since_process_start_s(), rss_gb());
  (void)t_rm;
  if (engine == "push") {
    const int rc = discovery_push(a, recs, bed, ctx, ss, rm, min_overlap, max_pair_distance, sample, flat_codes, flat_start, flat_size, env);
    // The outputs are written and closed: the process ends here.  Handing tens of gigabytes of host tables, the seqset, the
    // readmap and the contexts back piece by piece before exiting took most of a second of a ten-second command.
    cpuprof::stop_and_write();
    fflush(nullptr);
    if (getenv("BG_EXIT_PROBE")) {  // how long the device side of leaving takes, apart from the host's pages
      void* h = dlopen("libamdhip64.so", RTLD_NOLOAD | RTLD_NOW);
      if (!h) h = dlopen("libamdhip64.so.7", RTLD_NOLOAD | RTLD_NOW);
      int (*reset)() = h ? reinterpret_cast<int (*)()>(dlsym(h, "hipDeviceReset")) : nullptr;
      const double t0 = now_s();
      const int e = reset ? reset() : -1;
      fprintf(stderr, "discovery: hipDeviceReset -> %d in %.3f s; %.2f s since the process started\n", e, now_s() - t0, since_process_start_s());
    }
    // (BG_NORMAL_EXIT=1: through the exit handlers -- a profiler preloaded into the process writes its results there)
    if (const char* e = getenv("BG_NORMAL_EXIT")) if (e[0] == '1') exit(rc);
    _exit(rc);
  }
  std::string vcf = "##fileformat=VCFv4.1\n##source=biograph_amd graph_discover\n#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\n";
  uint64_t n_var = 0;
  for (auto& rec : recs) {
    const std::string& seq = rec.second;
    std::vector<int64_t> eo;
    std::vector<std::string> es;
    size_t run = 0;
    for (size_t i = 0; i <= seq.size(); ++i) {  // extents = maximal ACGT stretches (supercontigs)
      bool acgt = i < seq.size() && (seq[i] == 'A' || seq[i] == 'C' || seq[i] == 'G' || seq[i] == 'T');
      if (acgt) { ++run; continue; }
      if (run) { eo.push_back(int64_t(i - run)); es.push_back(seq.substr(i - run, run)); }
      run = 0;
    }
    if (es.empty()) continue;
    std::vector<const char*> ep;
    std::vector<uint64_t> el;
    for (auto& x : es) { ep.push_back(x.c_str()); el.push_back(x.size()); }
    bg_asm_list* l = nullptr;
    check(bg_asm_list_create(&l), "bg_asm_list_create");
    if (a.has("bed")) {
      auto it = bed.find(rec.first);
      if (it == bed.end()) { bg_asm_list_free(l); continue; }
      for (auto& iv : it->second)
        check(bg_make_ref_assemblies(l, es.size(), eo.data(), ep.data(), el.data(), iv.first, std::min<int64_t>(iv.second, int64_t(seq.size())), 100),
              "make_ref_assemblies");
      check(bg_asm_list_sort(l), "bg_asm_list_sort");
    } else {
      check(bg_make_ref_assemblies(l, es.size(), eo.data(), ep.data(), el.data(), 0, int64_t(seq.size()), 100), "make_ref_assemblies");
    }
    auto trim_and_update = [&]() {
      check(bg_graph_trim_ref(l, es.size(), eo.data(), ep.data(), el.data(), 0), "graph_trim_ref");
      check(bg_dedup_assemblies(l), "dedup");
      check(bg_update_rc_seqset_entries(ctx, ss, nullptr, l, 20), "update_rc_seqset_entries");
    };
    trim_and_update();
    check(bg_discover_branch(ctx, ss, rm, l, min_overlap, "BRANCH", ""), "discover_branch");
    trim_and_update();
    check(bg_discover_push_to_pair(ctx, ss, rm, l, min_overlap, max_pair_distance, "PUSH_TO_PAIR", "BRANCH"), "discover_push_to_pair");
    trim_and_update();
    for (uint64_t i = 0; i < bg_asm_list_size(l); ++i) {
      int64_t lo, ro;
      const char* sq;
      uint64_t ln;
      int mr;
      check(bg_asm_list_get(l, i, &lo, &ro, &sq, &ln, &mr, nullptr), "bg_asm_list_get");
      if (mr || lo == BG_OFFSET_NONE || ro == BG_OFFSET_NONE) continue;
      std::string r = seq.substr(size_t(lo), size_t(ro - lo)), alt(sq, ln);
      int64_t pos = lo + 1;
      if (r.empty() || alt.empty() || r.size() != alt.size()) {  // indel: pad with the preceding base
        if (lo == 0) continue;
        r = seq[size_t(lo) - 1] + r;
        alt = seq[size_t(lo) - 1] + alt;
        pos = lo;
      }
      vcf += rec.first + "\t" + std::to_string(pos) + "\t.\t" + r + "\t" + alt + "\t.\t.\t.\n";
      ++n_var;
    }
    bg_asm_list_free(l);
  }
  write_file(a.get("out"), vcf);
  fprintf(stderr, "%llu anchored variants over %zu scaffold(s), min overlap %u, %.2f s\n", (unsigned long long)n_var,
          recs.size(), min_overlap, now_s() - t0);
  bg_readmap_free(rm);
  bg_seqset_free(ss);
  bg_ctx_destroy(ctx);
  return 0;
}

int main(int argc, char** argv) {
  // The tracer sessions of `discovery` allocate and free hundreds of thousands of small search states per work item on 16
  // threads: with glibc's defaults the heap tops are trimmed and faulted in again all the time, and page faults of one
  // process serialise on its address-space lock (measured on the chr20-shaped sample: tracing 16.2 s -> 12.7 s).
  if (argc < 2) die("usage: bgbinary {create|merge|refmap|discovery} [options]   (MI355X build of the seqset path)");
  std::string cmd = argv[1];
  if (cmd == "bam2fastq") return cmd_bam2fastq(argc, argv);
  if (cmd == "create") return cmd_create(argc, argv);
  if (cmd == "refmap") return cmd_refmap(argc, argv);
  if (cmd == "reference") return cmd_reference(argc, argv);
  if (cmd == "merge") return cmd_merge(argc, argv);
  if (cmd == "discovery") return cmd_discovery(argc, argv);
  die("bgbinary: unknown command '" + cmd + "' (this build provides: create, merge, refmap, discovery, reference, bam2fastq)");
  return 1;
}
