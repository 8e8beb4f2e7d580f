import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print("ms_per_step", d["ms_per_step"])
x=d["discover"]
print("  ref_map %.1f ms  branch %.1f ms  propagate %.1f ms  push_to_pair %.1f ms  readmap %.1f ms" % (x["ref_map"]["seconds"]*1e3, x["branch_discover"]["seconds"]*1e3, x["push_to_pair_extension"]["propagate_seconds"]*1e3, x["push_to_pair_extension"]["trace_seconds"]*1e3, x["readmap"]["seconds"]*1e3))
e=d.get("end_to_end_from_fastq")
if e:
    print("  create", e["seconds"], "discovery", {k:e["discovery"][k] for k in ("process_wall","tracing_s","pipeline_s","coverage_genotype_s","vcf_records")})
