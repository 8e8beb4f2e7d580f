// bgbinary — command-line front end over libbgseq.so mirroring the reference's `bgbinary create`
// (modules/biograph/biograph_create.cpp:258-335 flags, :516-527 output layout) and the reference-map
// stage of `bgbinary discovery` (modules/biograph/biograph_variants.cpp:642-667).
//
//   bgbinary create --reads X.fq[.gz] [--pair Y.fq] [--interleaved] [--ref DIR] --out S.bg [--id ID] [-f]
//                   [--kmer-size 30] [--min-kmer-count 5] [--trim-after-portion 0.7]
//                   [--max-corrections 8] [--min-good-run 2] [--min-reads 0.4] [--warn-reads 0.7] [--stats FILE]
//   bgbinary refmap --in S.bg --ref DIR|FASTA --out FILE          (ref_map::build -> one byte per entry)
//   bgbinary discovery --in S.bg --ref DIR|FASTA --out V.vcf [--sample ID] [--min-overlap 0.7|N]   (graph_discover pipeline)
//   bgbinary merge --in A.bg --in B.bg [...] --out M.bg [--id ID] [-f]   (biograph_merge.cpp:189-400)
//
// Host work here is file I/O only (FASTQ parsing with zlib, output directory); every computation is a
// C-ABI call that runs on the GPU.  The sample's readmap lands in coverage/<sha1>.readmap and is listed
// under "samples" in metadata/bg_info.json (biograph_create.cpp:812-831, biograph_dir.cpp).
#include <errno.h>
#include <string.h>
#include <sys/stat.h>
#include <time.h>
#include <zlib.h>

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <string>
#include <vector>

#include "../../../include/bgseq.h"

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

static Args parse(int argc, char** argv, int first) {
  static const char* flags[] = {"interleaved", "force", "f", "allow-long-reads", nullptr};
  Args a;
  for (int i = first; i < argc; ++i) {
    std::string s = argv[i];
    if (s.rfind("--", 0) != 0 && s != "-f") die("unexpected argument: " + s);
    std::string key = s == "-f" ? "force" : s.substr(2), val;
    size_t eq = key.find('=');
    bool is_flag = false;
    for (const char** f = flags; *f; ++f) is_flag |= key == *f;
    if (eq != std::string::npos) { val = key.substr(eq + 1); key = key.substr(0, eq); }
    else if (!is_flag) { if (i + 1 >= argc) die("missing value for --" + key); val = argv[++i]; }
    if (key == "in") key = "reads";  // --reads,in (biograph_create.cpp:263)
    a.opt[key].push_back(val);
  }
  return a;
}

// FASTQ (plain or gzip) -> sequences; qualities dropped (read_importer.cpp:22)
static void read_fastq(const std::string& path, std::vector<std::string>* out) {
  gzFile f = gzopen(path == "-" ? "/dev/stdin" : path.c_str(), "rb");
  if (!f) die("Could not open " + path + ": " + strerror(errno));
  gzbuffer(f, 1 << 20);
  std::vector<char> buf(1 << 16);
  std::string line;
  uint64_t lineno = 0;
  auto getline = [&](std::string* l) -> bool {
    l->clear();
    for (;;) {
      if (!gzgets(f, buf.data(), int(buf.size()))) return !l->empty();
      l->append(buf.data());
      if (!l->empty() && l->back() == '\n') { l->pop_back(); if (!l->empty() && l->back() == '\r') l->pop_back(); return true; }
    }
  };
  while (getline(&line)) {
    if (lineno % 4 == 0 && !line.empty() && line[0] != '@') die(path + ": malformed FASTQ record at line " + std::to_string(lineno + 1));
    if (lineno % 4 == 1) out->push_back(line);
    ++lineno;
  }
  gzclose(f);
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

static int cmd_create(int argc, char** argv) {
  Args a = parse(argc, argv, 2);
  if (!a.has("reads") || !a.has("out")) die("create: --reads and --out are required");
  std::string out = a.get("out");
  if (out.size() < 3 || out.substr(out.size() - 3) != ".bg") out += ".bg";  // biograph_dir appends .bg
  if (exists(out + "/seqset") && !a.has("force")) die("Refusing to overwrite " + out + ". Use --force to override.");
  bg_create_params p;
  bg_create_params_default(&p);
  p.kmer_size = atoi(a.get("kmer-size", "30").c_str());
  p.min_kmer_count = uint32_t(atoi(a.get("min-kmer-count", "5").c_str()));
  p.correct.trim_after_portion = strtof(a.get("trim-after-portion", "0.7").c_str(), nullptr);
  p.correct.max_corrections = uint32_t(atoi(a.get("max-corrections", "8").c_str()));
  p.correct.min_good_run = uint32_t(atoi(a.get("min-good-run", "2").c_str()));
  double min_reads = atof(a.get("min-reads", "0.4").c_str()), warn_reads = atof(a.get("warn-reads", "0.7").c_str());
  if (p.min_kmer_count < 1) die("--min-kmer-count must be at least 1");

  double t0 = now_s();
  fprintf(stderr, "Importing reads\n");
  std::vector<std::string> reads;
  const auto& ins = a.opt["reads"];
  const auto& pairs = a.has("pair") ? a.opt["pair"] : std::vector<std::string>();
  if (!pairs.empty() && pairs.size() != ins.size()) die("--pair must be given once per --reads");
  for (size_t i = 0; i < ins.size(); ++i) {
    std::vector<std::string> r1, r2;
    read_fastq(ins[i], &r1);
    if (!pairs.empty()) {
      read_fastq(pairs[i], &r2);
      if (r1.size() != r2.size()) die("Pair files hold different numbers of reads");
      for (size_t j = 0; j < r1.size(); ++j) { reads.push_back(std::move(r1[j])); reads.push_back(std::move(r2[j])); }
    } else {
      for (auto& s : r1) reads.push_back(std::move(s));
    }
  }
  if (reads.empty()) die("\nNo reads were imported, exiting.");
  fprintf(stderr, "\nTotal reads imported: %zu\n", reads.size());
  double t_import = now_s() - t0;

  bg_ctx* ctx = nullptr;
  check(bg_ctx_create(0, nullptr, &ctx), "bg_ctx_create");
  std::string seq;
  std::vector<uint64_t> offs{0};
  size_t total = 0;
  for (auto& r : reads) total += r.size();
  seq.reserve(total);
  for (auto& r : reads) { seq += r; offs.push_back(seq.size()); }
  void *d_seq = nullptr, *d_offs = nullptr;
  check(bg_dev_alloc(ctx, seq.size(), &d_seq), "bg_dev_alloc");
  check(bg_dev_alloc(ctx, offs.size() * 8, &d_offs), "bg_dev_alloc");
  check(bg_dev_upload(ctx, d_seq, seq.data(), seq.size()), "upload");
  check(bg_dev_upload(ctx, d_offs, offs.data(), offs.size() * 8), "upload");
  fprintf(stderr, "\nRunning kmerization, read correction, seqset and readmap build on the GPU\n");
  double t1 = now_s();
  bg_seqset* ss = nullptr;
  bg_readmap* rm = nullptr;
  bg_build_stats st;
  check(bg_create_sample(ctx, static_cast<const uint8_t*>(d_seq), static_cast<const uint64_t*>(d_offs), reads.size(), &p,
                         pairs.empty() ? 0 : 1, &ss, &rm, &st),
        "bg_create_sample");
  double t_build = now_s() - t1;
  double pct = double(st.n_corrected_reads) / double(reads.size());
  char msg[256];
  if (pct < min_reads) {  // biograph_create.cpp:748-757
    snprintf(msg, sizeof msg, "Fewer than %2.0f%% of reads (set by --min-reads) were kept after correction (%llu / %zu remain). Cannot continue.",
             min_reads * 100.0, (unsigned long long)st.n_corrected_reads, reads.size());
    die(msg);
  }
  if (pct < warn_reads)
    fprintf(stderr, "Warning: Fewer than %2.0f%% of reads (set by --warn-reads) survived correction (%llu / %zu remain)\n",
            warn_reads * 100.0, (unsigned long long)st.n_corrected_reads, reads.size());

  mkdirs(out);
  mkdirs(out + "/qc");
  mkdirs(out + "/metadata");
  mkdirs(out + "/coverage");
  std::vector<const char*> av(argv, argv + argc);
  check(bg_seqset_write(ctx, ss, (out + "/seqset").c_str(), argc, av.data()), "bg_seqset_write");
  char sha[41];
  std::string rm_tmp = out + "/coverage/tmp.readmap";
  check(bg_readmap_write(ctx, rm, rm_tmp.c_str(), argc, av.data(), sha), "bg_readmap_write");
  if (rename(rm_tmp.c_str(), (out + "/coverage/" + sha + ".readmap").c_str()) != 0) die("Could not rename " + rm_tmp);
  double t_total = now_s() - t0;

  std::string id = a.get("id");
  char js[2048];
  snprintf(js, sizeof js,
           "{\"command\":\"create\",\"version\":\"biograph_amd-r1\",\"accession_id\":\"%s\",\"imported_reads\":%zu,"
           "\"corrected_reads\":%llu,\"corrected_bases\":%llu,\"avg_bases_per_read\":%.6f,\"corrected_pct\":%.6f,"
           "\"kept_kmers\":%llu,\"seqset_entries\":%llu,\"timings\":[{\"import\":%.0f},{\"gpu_build\":%.3f},{\"total\":%.0f}]}",
           id.c_str(), reads.size(), (unsigned long long)st.n_corrected_reads, (unsigned long long)st.n_corrected_bases,
           st.n_corrected_reads ? double(st.n_corrected_bases) / double(st.n_corrected_reads) : 0.0, pct,
           (unsigned long long)st.n_kept_kmers, (unsigned long long)st.n_entries, t_import, t_build, t_total);
  write_file(out + "/qc/create_stats.json", js);
  if (a.has("stats")) write_file(a.get("stats"), js);
  snprintf(js, sizeof js, "{\"accession_id\":\"%s\",\"command_history\":[],\"samples\":{\"%s\":\"%s\"},\"version\":\"biograph_amd-r1\"}",
           id.c_str(), id.c_str(), sha);
  write_file(out + "/metadata/bg_info.json", js);
  fprintf(stderr, "%llu / %zu reads survived read correction; %llu seqset entries; GPU build %.3f s\n",
          (unsigned long long)st.n_corrected_reads, reads.size(), (unsigned long long)st.n_entries, t_build);
  bg_readmap_free(rm);
  bg_seqset_free(ss);
  bg_dev_free(ctx, d_seq);
  bg_dev_free(ctx, d_offs);
  bg_ctx_destroy(ctx);
  return 0;
}

static int cmd_refmap(int argc, char** argv) {
  Args a = parse(argc, argv, 2);
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
static std::string json_string(const std::string& js, const std::string& key) {
  std::string pat = "\"" + key + "\":\"";
  size_t p = js.find(pat);
  if (p == std::string::npos) return "";
  p += pat.size();
  return js.substr(p, js.find('"', p) - p);
}
static std::map<std::string, std::string> json_samples(const std::string& js) {
  std::map<std::string, std::string> out;
  size_t p = js.find("\"samples\":{");
  if (p == std::string::npos) return out;
  p += 11;
  size_t end = js.find('}', p);
  while (p < end) {
    size_t k0 = js.find('"', p);
    if (k0 == std::string::npos || k0 >= end) break;
    size_t k1 = js.find('"', k0 + 1), v0 = js.find('"', k1 + 1), v1 = js.find('"', v0 + 1);
    out[js.substr(k0 + 1, k1 - k0 - 1)] = js.substr(v0 + 1, v1 - v0 - 1);
    p = v1 + 1;
  }
  return out;
}

// bgbinary merge --in A.bg --in B.bg [...] --out M.bg [--id ID] [-f]   (biograph_merge.cpp:189-400)
static int cmd_merge(int argc, char** argv) {
  Args a = parse(argc, argv, 2);
  if (!a.has("reads") || !a.has("out")) die("merge: --in (at least once) and --out are required");
  const auto& ins = a.opt["reads"];  // the parser files --in under --reads
  std::string out = a.get("out");
  if (out.size() < 3 || out.substr(out.size() - 3) != ".bg") out += ".bg";
  if (exists(out + "/seqset") && !a.has("force")) die("Refusing to overwrite " + out + ". Use --force to override.");
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
  for (size_t i = 0; i < ins.size(); ++i) {
    for (auto& kv : json_samples(infos[i])) {
      bg_readmap *rm = nullptr, *mig = nullptr;
      check(bg_readmap_open(ctx, (ins[i] + "/coverage/" + kv.second + ".readmap").c_str(), &rm), "bg_readmap_open");
      check(bg_readmap_migrate(ctx, rm, mapp[i], merged, &mig), "bg_readmap_migrate");
      char sha[41];
      std::string tmp = out + "/coverage/tmp.readmap";
      check(bg_readmap_write(ctx, mig, tmp.c_str(), argc, av.data(), sha), "bg_readmap_write");
      if (rename(tmp.c_str(), (out + "/coverage/" + sha + ".readmap").c_str()) != 0) die("Could not rename " + tmp);
      samples_json += std::string(n_samples ? "," : "") + "\"" + kv.first + "\":\"" + sha + "\"";
      ++n_samples;
      bg_readmap_free(mig);
      bg_readmap_free(rm);
    }
  }
  std::string meta = "{\"accession_id\":\"" + merged_id + "\",\"command_history\":[],\"samples\":{" + samples_json +
                     "},\"version\":\"biograph_amd-r1\"}";
  write_file(out + "/metadata/bg_info.json", meta);
  char js[1024];
  snprintf(js, sizeof js, "{\"command\":\"merge\",\"version\":\"biograph_amd-r1\",\"accession_id\":\"%s\",\"samples\":%zu,"
           "\"seqset_entries\":%llu}", merged_id.c_str(), n_samples, (unsigned long long)bg_seqset_size(merged));
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
static int cmd_discovery(int argc, char** argv) {
  Args a = parse(argc, argv, 2);
  if (!a.has("reads") || !a.has("ref") || !a.has("out")) die("discovery: --in S.bg, --ref and --out are required");
  std::string bgdir = a.get("reads");
  // scaffolds (chromosomes) and their N-free supercontigs from the reference directory / .ref / FASTA; a
  // scaffold is rebuilt with N between its supercontigs so that VCF positions are scaffold coordinates
  bg_reference* rf = nullptr;
  check(bg_reference_open(a.get("ref").c_str(), &rf), "bg_reference_open");
  std::vector<std::pair<std::string, std::string>> recs;
  for (uint32_t c = 0; c < bg_reference_scaffold_count(rf); ++c)
    recs.emplace_back(bg_reference_scaffold_name(rf, c), std::string(bg_reference_scaffold_len(rf, c), 'N'));
  for (uint32_t i = 0; i < bg_reference_supercontig_count(rf); ++i) {
    uint32_t sc = 0;
    uint64_t off = 0, len = 0, flat = 0;
    check(bg_reference_supercontig(rf, i, &sc, &off, &len, &flat), "bg_reference_supercontig");
    if (off + len > recs[sc].second.size()) recs[sc].second.resize(off + len, 'N');
    check(bg_reference_bases(rf, flat, len, &recs[sc].second[off]), "bg_reference_bases");
  }
  bg_reference_free(rf);
  bg_ctx* ctx = nullptr;
  check(bg_ctx_create(0, nullptr, &ctx), "bg_ctx_create");
  bg_seqset* ss = nullptr;
  check(bg_seqset_open(ctx, (bgdir + "/seqset").c_str(), &ss), "bg_seqset_open");
  auto samples = json_samples(read_text(bgdir + "/metadata/bg_info.json"));
  std::string sample = a.get("sample");
  if (samples.empty()) die("No samples found in BioGraph metadata.");                           // biograph_dir.cpp:104-107
  if (sample.empty()) {
    if (samples.size() > 1) die("Multiple samples are present. You must specify a readmap or accession ID.");  // :108-111
    sample = samples.begin()->first;
  }
  if (!samples.count(sample)) die("Sample " + sample + " not found in " + bgdir);
  bg_readmap* rm = nullptr;
  check(bg_readmap_open(ctx, (bgdir + "/coverage/" + samples[sample] + ".readmap").c_str(), &rm), "bg_readmap_open");
  double mo = atof(a.get("min-overlap", "0.7").c_str());
  uint32_t min_overlap = mo < 1.0 ? uint32_t(mo * bg_readmap_max_read_len(rm)) : uint32_t(mo);  // biograph_variants.cpp:674
  int64_t max_pair_distance = atoll(a.get("max-pair-distance", "1000").c_str());
  double t0 = now_s();
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
    check(bg_make_ref_assemblies(l, es.size(), eo.data(), ep.data(), el.data(), 0, int64_t(seq.size()), 100), "make_ref_assemblies");
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
  if (argc < 2) die("usage: bgbinary {create|merge|refmap|discovery} [options]   (MI355X build of the seqset path)");
  std::string cmd = argv[1];
  if (cmd == "create") return cmd_create(argc, argv);
  if (cmd == "refmap") return cmd_refmap(argc, argv);
  if (cmd == "merge") return cmd_merge(argc, argv);
  if (cmd == "discovery") return cmd_discovery(argc, argv);
  die("bgbinary: unknown command '" + cmd + "' (this build provides: create, merge, refmap, discovery)");
  return 1;
}
