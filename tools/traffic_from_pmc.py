"""profiles/traffic.json from the per-case PMC CSVs of tools/profile_round3.sh:
HBM bytes per launch of the dominant (most dispatched rollout) kernel = (2 x FETCH_SIZE + WRITE_SIZE) x 1024
(FETCH_SIZE reads half the bytes on gfx950, WRITE_SIZE is exact: profiles/README.md, calibration CSVs).
    python tools/traffic_from_pmc.py <dir with <case>_fetch.csv / <case>_write.csv> <tag> <commit>"""
import csv, json, os, sys
d, tag, commit = sys.argv[1:4]
KEYS = {"B8": "rollout_philox_B8", "B1": "rollout_philox_B1", "B1_lean": "rollout_philox_B1_lean", "B64": "rollout_philox_B64", "B64_lean": "rollout_philox_B64_lean",
        "B256": "rollout_wave_philox_B256", "B256_lean": "rollout_wave_philox_B256_lean", "sampled": "rollout_sampled_K8192",
        "c5": "rollout_K16384_T100_G512", "ref5000": "rollout_K5000_T50_G64"}


def dominant(path, counter):
    rows = [r for r in csv.DictReader(open(path)) if "rollout" in r["kernel"] and r["counter"] == counter]
    r = max(rows, key=lambda r: int(r["dispatches"]))
    return float(r["avg"]), r["kernel"].split("(")[0][-60:], int(r["dispatches"])


out, detail = {}, {}
for case, key in KEYS.items():
    f, w = os.path.join(d, f"{case}_fetch.csv"), os.path.join(d, f"{case}_write.csv")
    if not (os.path.exists(f) and os.path.exists(w)):
        continue
    fk, kf, nf = dominant(f, "FETCH_SIZE")
    wk, kw, nw = dominant(w, "WRITE_SIZE")
    out[key] = int(round((2 * fk + wk) * 1024))
    detail[case] = {"fetch_KiB_counter": fk, "fetch_KiB_corrected": 2 * fk, "write_KiB": wk, "kernel": kf, "dispatches": [nf, nw]}
out["collected"] = f"profiles/{tag}_pmc/*_fetch.csv + *_write.csv at commit {commit}: (2 x FETCH_SIZE + WRITE_SIZE) KiB x 1024 per launch, one bench leg per rocprofv3 --pmc pass (tools/profile_round6.sh)"
out["source"] = out["collected"]
out["detail"] = detail
json.dump(out, open(os.path.join(d, "traffic.json"), "w"), indent=1)
print(json.dumps({k: v for k, v in out.items() if k not in ("detail",)}, indent=1))
