import sys, json
tag=None
for line in open(sys.argv[1]):
    line=line.strip()
    if line.startswith("=="): tag=line
    elif line.startswith("{"):
        d=json.loads(line); pk=(d.get("roofline") or {}).get("per_kernel") or {}   # --no_profile lines carry no kernel table
        x=[v for k,v in pk.items() if "cross-attention" in k]
        print(tag, d["ms_per_step"], "table", (d.get("roofline") or {}).get("table_pass_ms_per_step"), "xattn", x[0]["ms_per_step"] if x else None, "verified", d.get("verified"), d.get("psnr_fast_vs_plain_kernels_db"))
