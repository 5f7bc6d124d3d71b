#!/usr/bin/env python
"""Merge three rocprofv3 --pmc passes (SQ+GRBM | FETCH_SIZE | WRITE_SIZE) of tools/prof_step.py into one
per-kernel table and a per-kernel-class traffic summary (HBM bytes per launch, FETCH_SIZE doubled as
MI355X_MICROARCH.md prescribes for 16 B/lane streaming reads on gfx950).
usage: pmc_table.py <dir_sq> <dir_fetch> <dir_write> <out_csv> <out_json> [precision label] [workload label]
Per class the JSON also carries the shader clock the kernels of that class ran at (GRBM_GUI_ACTIVE / 8 XCDs / duration).
"""
import collections, csv, glob, json, sys

def load(d):
    import os
    f = max(glob.glob(d + "/*/*counter_collection.csv"), key=os.path.getmtime)     # gpurun merges into a directory that may hold older runs
    out = collections.OrderedDict()
    for r in csv.DictReader(open(f)):
        e = out.setdefault(r["Dispatch_Id"], {"name": r["Kernel_Name"], "grid": r["Grid_Size"],
                                               "dt": (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3})
        e[r["Counter_Name"]] = float(r["Counter_Value"])
    rows = list(out.values())
    idx = [i for i, e in enumerate(rows) if "logmel" in e["name"]]
    return rows[idx[-1]:]           # the last forward only

def cls(name):
    for key, c in (("logmel", "frontend"), ("stem_kernel", "stem"), ("dwconv7", "dwconv"), ("mlp_fused_wide", "mlp_wide"), ("mlp_fused_stat", "mlp_wide"), ("mlp_fused", "mlp_fused"),
                   ("rowstats", "rowstats"), ("pool_head", "poolhead"), ("nhwc_to_nchw", "transpose")):
        if key in name:
            return c
    if "gemm_split16_kernel" in name:          # <EPI, GATHER>
        t = name.split("<")[1].split(">")[0].replace(" ", "").split(",")
        return {"1": "pw1", "2": "pw2"}.get(t[0], "downsample")
    if "gemm_f32_kernel" in name or "gemm_split_kernel" in name or "gemm_bf16_kernel" in name:
        t = name.split("<")[1].split(">")[0].replace(" ", "").split(",")
        return {"1": "pw1", "2": "pw2"}.get(t[4], "downsample")
    return None

def traffic_summary(fe, wr, sq=None):
    """{class: {launches_per_step, hbm_traffic_bytes_per_launch, fetch_bytes_per_launch_x2, write_bytes_per_launch[, shader_clock_GHz,
    mfma_util_cycles]}} from the FETCH_SIZE and WRITE_SIZE passes (load()), and the SQ + GRBM pass when given.  A pass that carries
    GRBM_GUI_ACTIVE beside its own counter also yields the clock."""
    agg = collections.OrderedDict()
    for i, (y, z) in enumerate(zip(fe, wr)):
        c = cls(y["name"])
        if c is None:
            continue
        a = agg.setdefault(c, {"launches": 0, "fetch_MB_x2": 0.0, "write_MB": 0.0, "mfma_busy": 0.0, "simd_cycles": 0.0, "gui": 0.0, "dt_us": 0.0})
        a["launches"] += 1
        a["fetch_MB_x2"] += 2 * y["FETCH_SIZE"] * 1024 / 1e6
        a["write_MB"] += z["WRITE_SIZE"] * 1024 / 1e6
        x = sq[i] if sq is not None else (z if "GRBM_GUI_ACTIVE" in z else None)
        if x is not None:
            gui = x["GRBM_GUI_ACTIVE"] / 8
            a["gui"] += gui; a["dt_us"] += x["dt"]
            a["mfma_busy"] += x.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0); a["simd_cycles"] += gui * 1024
    out = {}
    for c, a in agg.items():
        e = {"launches_per_step": a["launches"], "hbm_traffic_bytes_per_launch": (a["fetch_MB_x2"] + a["write_MB"]) * 1e6 / a["launches"],
             "fetch_bytes_per_launch_x2": a["fetch_MB_x2"] * 1e6 / a["launches"], "write_bytes_per_launch": a["write_MB"] * 1e6 / a["launches"]}
        if a["dt_us"] > 0:
            e["shader_clock_GHz"] = a["gui"] / a["dt_us"] / 1e3
            if sq is not None:
                e["mfma_util_cycles"] = a["mfma_busy"] / a["simd_cycles"]
        out[c] = e
    return out


def main():
    PREC = sys.argv[6] if len(sys.argv) > 6 else "fp32_split"
    WORK = sys.argv[7] if len(sys.argv) > 7 else "one forward, B=64, 10 s @ 32 kHz"
    sq, fe, wr = load(sys.argv[1]), load(sys.argv[2]), load(sys.argv[3])
    with open(sys.argv[4], "w") as out:
        out.write("# %s (tools/prof_step.py --precision %s, one stream); three separate rocprofv3 --pmc passes.\n" % (WORK, PREC))
        out.write("# fetch_MB_x2 = 2 * FETCH_SIZE (gfx950 reports half the bytes of 16 B/lane streaming reads); write_MB = WRITE_SIZE.\n")
        out.write("class,kernel,grid,dur_us,clock_GHz,waves_per_SIMD,wait_any,wait_inst_any,valu_active,mfma_util,lds_bank_conflict_per_cu_cycle,fetch_MB_x2,write_MB\n")
        for x, y, z in zip(sq, fe, wr):
            c = cls(x["name"])
            if c is None:
                continue
            gui = x["GRBM_GUI_ACTIVE"] / 8
            fetch, write = 2 * y["FETCH_SIZE"] * 1024 / 1e6, z["WRITE_SIZE"] * 1024 / 1e6
            mf = x.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (gui * 1024)
            out.write('%s,"%s",%s,%.1f,%.2f,%.2f,%.3f,%.3f,%.3f,%.3f,%.3f,%.1f,%.1f\n' % (
                c, x["name"][:70], x["grid"], x["dt"], gui / x["dt"] / 1e3, x["SQ_WAVE_CYCLES"] * 4 / (gui * 1024),
                x["SQ_WAIT_ANY"] / x["SQ_WAVE_CYCLES"], x["SQ_WAIT_INST_ANY"] / x["SQ_WAVE_CYCLES"],
                x["SQ_ACTIVE_INST_VALU"] / x["SQ_WAVE_CYCLES"], mf, x["SQ_LDS_BANK_CONFLICT"] / (gui * 256), fetch, write))
    summary = traffic_summary(fe, wr, sq)
    json.dump({"source": "rocprofv3 --pmc, 3 passes: SQ_*+GRBM_GUI_ACTIVE | FETCH_SIZE | WRITE_SIZE; FETCH_SIZE x2 (gfx950 correction)",
               "workload": "%s (tools/prof_step.py --precision %s, one stream)" % (WORK, PREC), "classes": summary}, open(sys.argv[5], "w"), indent=1)
    print(json.dumps(summary, indent=1))


if __name__ == "__main__":
    main()
