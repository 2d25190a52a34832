#!/usr/bin/env python
"""Memory-side traffic per launch of the hot kernels from a tools/pmc_probe.py summary (groups rdsize + wrsize): bytes from the request
counts BY SIZE (32 / 64 / 128 B reads, 32 / 64 B writes), next to the raw FETCH_SIZE-style figure (every request tallied at 64 B) and the
guide's x2 correction of it.  usage: pmc_traffic_json.py <pmc_summary.txt> <out.json>"""
import json
import re
import sys


def parse(path):
    d, cur = {}, None
    for line in open(path):
        m = re.match(r"\s+(\S+)\s+avg/dispatch =\s+([0-9.]+)", line)
        if m and cur:
            d.setdefault(cur, {})[m.group(1)] = float(m.group(2))
        elif line.strip() and not line.startswith("==") and not line.startswith(" "):
            k = line.strip()
            k = re.sub(r"^void ", "", k); k = re.sub(r"[<(].*", "", k).replace("ngp::", "")
            cur = k
    return d


def main():
    d = parse(sys.argv[1])
    out = {"_comment": "memory-side (L2 <-> fabric / Infinity Cache / HBM) traffic per launch at the trained steady state (1000 pre-training steps), rocprofv3 --pmc, separate "
                       "passes per counter group (tools/pmc_probe.py rdsize / wrsize).  bytes = 32 n32 + 64 n64 + 128 n128 for reads (sizes counted by the TCC_EA0_RDREQ_* "
                       "counters), 64 n64 + 32 (n - n64) for writes; `as_fetch_size` = every request at 64 B (what FETCH_SIZE / WRITE_SIZE report), `fetch_x2` = the "
                       "guide's correction of FETCH_SIZE for wide streaming reads (MI355X_MICROARCH.md, HBM).  Infinity-Cache hits are included in all of them."}
    for k, c in d.items():
        if "TCC_EA0_RDREQ_sum" not in c and "TCC_EA0_WRREQ_sum" not in c:
            continue
        e = out.setdefault(k, {})
        if "TCC_EA0_RDREQ_sum" in c:
            n, n32, n64, n128 = c["TCC_EA0_RDREQ_sum"], c.get("TCC_EA0_RDREQ_32B_sum", 0), c.get("TCC_EA0_RDREQ_64B_sum", 0), c.get("TCC_EA0_RDREQ_128B_sum", 0)
            rest = max(n - n32 - n64 - n128, 0)
            e["read_requests"] = {"total": n, "32B": n32, "64B": n64, "128B": n128}
            e["read_bytes"] = 32 * n32 + 64 * (n64 + rest) + 128 * n128
            e["read_bytes_as_fetch_size"] = 64 * n
        if "TCC_EA0_WRREQ_sum" in c:
            n, n64 = c["TCC_EA0_WRREQ_sum"], c.get("TCC_EA0_WRREQ_64B_sum", 0)
            e["write_requests"] = {"total": n, "64B": n64, "atomics": c.get("TCC_EA0_ATOMIC_sum", 0)}
            e["write_bytes"] = 64 * n64 + 32 * max(n - n64, 0)
            e["write_bytes_as_write_size"] = 64 * n
    unit = ["k_train_fused" if "k_train_fused" in out else "k_train_fwd_bwd", "k_grad_bin", "k_grad_accumulate"]  # round 5: k_train_fused (T1 + W in one kernel) is the unit's front
    if all(u in out and "read_bytes" in out[u] and "write_bytes" in out[u] for u in unit):
        out["k_train_fwd_bwd+k_grad_bin+k_grad_accumulate"] = {
            "kernels": unit,
            "bytes_per_launch": int(sum(out[u]["read_bytes"] + out[u]["write_bytes"] for u in unit)),
            "bytes_per_launch_as_fetch_write_size": int(sum(out[u]["read_bytes_as_fetch_size"] + out[u]["write_bytes_as_write_size"] for u in unit)),
            "bytes_per_launch_fetch_x2": int(sum(2 * out[u]["read_bytes_as_fetch_size"] + out[u]["write_bytes_as_write_size"] for u in unit))}
    for k in ("k_inference_tiles", "k_optimizer"):
        if k in out and "read_bytes" in out[k] and "write_bytes" in out[k]:
            out[k]["bytes_per_launch"] = int(out[k]["read_bytes"] + out[k]["write_bytes"])
    if "k_inference_tiles" in out:
        out["k_inference"] = dict(out["k_inference_tiles"], note="per k_inference_tiles launch (3 per step)")
    json.dump(out, open(sys.argv[2], "w"), indent=1)
    print(json.dumps({k: v.get("bytes_per_launch") for k, v in out.items() if isinstance(v, dict) and "bytes_per_launch" in v}))


if __name__ == "__main__":
    main()
