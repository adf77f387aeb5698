"""Reads the rocprofv3 JSON of tools/placement_channels.sh: per dispatch of k_probe_write and per counter, the values by dimension (XCC, TCC instance).
Prints, for the fast-block and the slow-block dispatches of tools/placement_pmc.py's final sequence (6 fast, 6 slow, 6 fast), the per-instance means and
how concentrated they are (max / mean, instances above 1.5 x mean).  Tools only; written to cope with whatever layout the JSON has."""
import glob, gzip, json, os, sys
import numpy as np

def load(path):
    op = gzip.open if path.endswith(".gz") else open
    with op(path, "rt") as f:
        return json.load(f)

def walk(o, key):
    if isinstance(o, dict):
        for k, v in o.items():
            if k == key:
                yield v
            yield from walk(v, key)
    elif isinstance(o, list):
        for v in o:
            yield from walk(v, key)

out = {}
for d in sorted(glob.glob(os.path.join(sys.argv[1], "[0-9]"))):
    files = glob.glob(os.path.join(d, "**", "*results.json*"), recursive=True)
    if not files:
        out[os.path.basename(d)] = "no results.json"
        continue
    js = load(files[0])
    tool = js["rocprofiler-sdk-tool"][0] if "rocprofiler-sdk-tool" in js else js
    # counter id -> name, dimension tables
    names = {}
    for c in walk(tool, "counters"):
        if isinstance(c, list):
            for e in c:
                if isinstance(e, dict) and "id" in e and "name" in e:
                    names[e["id"]["handle"] if isinstance(e["id"], dict) else e["id"]] = e
    kern = {}
    for ks in walk(tool, "kernel_symbols"):
        if isinstance(ks, list):
            for e in ks:
                if isinstance(e, dict) and "kernel_id" in e:
                    kern[e["kernel_id"]] = e.get("formatted_kernel_name") or e.get("kernel_name")
    cc = None
    for v in walk(tool, "counter_collection"):
        if isinstance(v, list) and v:
            cc = v
            break
    if cc is None:
        out[os.path.basename(d)] = {"error": "no counter_collection", "top": list(tool.keys())[:20]}
        continue
    rows = []  # (dispatch order, kernel name, {dimension-key: value})
    for rec in cc:
        di = rec.get("dispatch_data", {}).get("dispatch_info", {})
        kname = kern.get(di.get("kernel_id"), "?")
        vals = {}
        for pos, r in enumerate(rec.get("records", [])):  # one record per (TCC instance, XCC), in the order of the counter's `instances` table
            cid = r.get("counter_id", {})
            cid = cid.get("handle") if isinstance(cid, dict) else cid
            vals[(cid, pos)] = r.get("value", r.get("counter_value"))
        rows.append((di.get("dispatch_id", len(rows)), kname, vals))
    rows.sort(key=lambda r: r[0])
    probe = [r for r in rows if "k_probe_write" in str(r[1])]
    res = {"dispatches": len(rows), "probe_dispatches": len(probe), "counter_names": {str(k): v.get("name") for k, v in names.items()}, "sample_record_keys": list(cc[0].get("records", [{}])[0].keys()) if cc[0].get("records") else []}
    if len(probe) >= 18:
        last = probe[-18:]
        for tag, sel in (("fast", last[:6] + last[12:]), ("slow", last[6:12])):
            keys = sorted(sel[0][2].keys())
            m = np.array([[float(r[2].get(k, 0) or 0) for k in keys] for r in sel]).mean(0)
            per_xcc = m.reshape(-1, 16).sum(1) if m.size % 16 == 0 else m
            res[tag] = {"per_xcc_sum": [round(float(x), 1) for x in per_xcc], "instances": len(keys), "sum": float(m.sum()), "mean": float(m.mean()), "max": float(m.max()), "min": float(m.min()), "max_over_mean": float(m.max() / max(m.mean(), 1e-9)),
                        "instances_above_1.5x_mean": int((m > 1.5 * m.mean()).sum()), "instances_below_0.5x_mean": int((m < 0.5 * m.mean()).sum()), "per_instance": [round(float(x), 1) for x in m]}
    out[os.path.basename(d)] = res
print(json.dumps(out))
