import re, sys
tot = {}
n = 0
for line in open(sys.argv[1]):
    if line.startswith("bg_tracer_trace:"):
        n += 1
        m = re.search(r"([\d.]+) s in all \(extents ([\d.]+), chunk ([\d.]+), anchor drop ([\d.]+), pair counter \+ reverse ([\d.]+)\): tables ([\d.]+) s, start ([\d.]+) s, device rounds ([\d.]+) s, host replay ([\d.]+) s, teardown ([\d.]+) s", line)
        if m:
            for k, v in zip(["all", "extents", "chunk", "anchor_drop", "pair_counter", "tables", "start", "device_rounds", "host_replay", "teardown"], m.groups()):
                tot[k] = tot.get(k, 0) + float(v)
    elif line.startswith("bg_pop_trace:"):
        m = re.search(r"anchor drops ([\d.]+) s, reference ([\d.]+) s .* assemble ([\d.]+) s", line)
        if m:
            for k, v in zip(["pop_drops", "pop_reference", "pop_assemble"], m.groups()):
                tot[k] = tot.get(k, 0) + float(v)
        m = re.search(r"reference: walk ([\d.]+) replay ([\d.]+) kept ([\d.]+) prefix reads ([\d.]+) items ([\d.]+) materialize ([\d.]+)", line)
        if m:
            for k, v in zip(["ref_walk", "ref_replay", "ref_kept", "ref_prefix_reads", "ref_items", "ref_materialize"], m.groups()):
                tot[k] = tot.get(k, 0) + float(v)
print(n, "push calls;", {k: round(v, 2) for k, v in tot.items()})
