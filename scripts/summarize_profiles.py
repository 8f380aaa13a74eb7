"""Copies the rocprofv3 summaries a gpurun call of scripts/gpu_profile_round.sh left under gpurun_out/ into profiles/
(tracked) and re-derives from them, without bench.py's own event brackets, every fraction bench.py's JSON line carries:

  roofline            k_kde_split: ALGORITHMIC flops K K' (3 P + 1) / rocprofv3 kernel duration against the dense 16-bit MFMA peak
                      (frac, frac_algorithmic), the f16 / bf16 MFMA work actually issued beside it (mfma_issue_frac), the
                      matrix-pipe busy fraction and the vector-issue fraction from the PMC pass, the clock the kernel ran at
  roofline_hbm        k_gram: algorithmic bytes / rocprofv3 kernel duration, PMC traffic beside it
  roofline_streaming  SURVEY 8(d): algorithmic bytes of a generation / (bench step time - pair-sum kernel), and set 0

HBM traffic = the separate FETCH_SIZE / WRITE_SIZE PMC passes with the gfx950 correction of
/opt/skills/guides/MI355X_MICROARCH.md (FETCH_SIZE reports 1/2 of a wide coalesced read; WRITE_SIZE exact).
usage: python scripts/summarize_profiles.py <round-tag> <stats-prefix> <pmc-prefix>      (e.g. r02 prof pmc)"""
import csv, glob, json, os, re, shutil, sys

tag, sp, pp = sys.argv[1], sys.argv[2], sys.argv[3]
G = 'gpurun_out/'
HBM_PEAK, MFMA_PEAK_TF, NSIMD = 8000.0, 2500.0, 1024


def find(d, suffix):
    return sorted(glob.glob(d + '/**/*' + suffix, recursive=True))[0]


def short(k):
    m = re.search(r'(k_[a-z0-9_]+(<[^>]*>)?)', k)
    return m.group(1) if m else k.split('(')[0].strip()


def pmc(d, counter=None, kernel=None):
    """per-kernel (or one kernel's per-counter) per-dispatch averages of a counter_collection.csv"""
    out = {}
    for r in csv.DictReader(open(find(d, 'counter_collection.csv'))):
        if counter is not None and r['Counter_Name'] != counter:
            continue
        if kernel is not None and kernel not in r['Kernel_Name']:
            continue
        key = r['Kernel_Name'] if counter is not None else r['Counter_Name']
        out.setdefault(key, []).append(float(r['Counter_Value']))
    return {k: sum(v) / len(v) for k, v in out.items()}


def stats(cfg):
    """kernel -> (calls, average ns) from the --kernel-trace --stats pass"""
    return {short(r['Name']): (int(r['Calls']), float(r['AverageNs'])) for r in csv.DictReader(open(find(G + '%s_c%d' % (sp, cfg), 'kernel_stats.csv')))}


def bench(cfg):
    return json.loads(open(G + 'bench_c%d.json' % cfg).read().strip().splitlines()[-1])


# ---- copies ---------------------------------------------------------------------------------------------------------------
cfgs = [c for c in (2, 3, 4, 5) if os.path.exists(G + 'bench_c%d.json' % c)]
for c in cfgs:
    shutil.copy(find(G + '%s_c%d' % (sp, c), 'kernel_stats.csv'), 'profiles/%s_kernel_stats_config%d.csv' % (tag, c))
    open('profiles/%s_bench_config%d.json' % (tag, c), 'w').write(open(G + 'bench_c%d.json' % c).read().strip().splitlines()[-1] + '\n')   # (the JSON line: librccl prints a banner in front of it)

# ---- HBM traffic per kernel -------------------------------------------------------------------------------------------------
res = {"_note": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in SEPARATE passes of `python3 bench.py --config N --steps 2 "
                "--warmup 1 --no-cpu-baseline`; per-dispatch averages in KiB; hbm_bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024 "
                "(gfx950: FETCH_SIZE counts 64 B per 128-B request).", "configs": {}}
rows = []
for cfg in (2, 3, 4, 5):
    if not os.path.isdir(G + '%s_fetch_c%d' % (pp, cfg)) or not os.path.isdir(G + '%s_write_c%d' % (pp, cfg)):
        continue
    fe, wr = pmc(G + '%s_fetch_c%d' % (pp, cfg), 'FETCH_SIZE'), pmc(G + '%s_write_c%d' % (pp, cfg), 'WRITE_SIZE')
    ent = {}
    for k in fe:
        hbm = (2 * fe[k] + wr.get(k, 0)) * 1024
        rows.append((cfg, short(k), fe[k], wr.get(k, 0), hbm))
        ent[short(k)] = {"FETCH_SIZE_KiB_raw": fe[k], "WRITE_SIZE_KiB": wr.get(k, 0), "hbm_bytes_per_launch": hbm}
    res["configs"][str(cfg)] = ent
json.dump(res, open('profiles/%s_pmc_hbm_traffic.json' % tag, 'w'), indent=1)
with open('profiles/%s_pmc_hbm_traffic.csv' % tag, 'w') as f:
    f.write("config,kernel,FETCH_SIZE_KiB_raw,WRITE_SIZE_KiB,hbm_bytes_corrected\n")
    for r in rows:
        f.write('%d,"%s",%.1f,%.1f,%.0f\n' % r)


def traffic(cfg, prefix):
    for k, v in res["configs"].get(str(cfg), {}).items():
        if k.startswith(prefix) and v["hbm_bytes_per_launch"] > 1e6:
            return v["hbm_bytes_per_launch"]
    return None


# ---- SQ counters of the Gram kernel -------------------------------------------------------------------------------------------
sq = {}
for d in (G + '%s_sq_c3' % pp, G + '%s_sq2_c3' % pp):
    if os.path.isdir(d):
        for kern in ('k_gram<3', 'k_gram_dma<3'):
            sq.update(pmc(d, kernel=kern))
if sq:
    sq['_note'] = ("rocprofv3 --pmc (two passes) on `python3 bench.py --config 3 --steps 2 --warmup 1 --no-cpu-baseline`, per-dispatch "
                   "averages of the k_gram<3,1> launch; MfmaUtil = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE/8 * 1024 SIMDs)")
    if 'GRBM_GUI_ACTIVE' in sq and 'SQ_VALU_MFMA_BUSY_CYCLES' in sq:
        sq['MfmaUtil_percent'] = 100 * sq['SQ_VALU_MFMA_BUSY_CYCLES'] / (sq['GRBM_GUI_ACTIVE'] / 8 * NSIMD)
    if 'SQ_WAIT_INST_ANY' in sq and 'SQ_WAVE_CYCLES' in sq:
        sq['wait_fraction_of_wave_cycles'] = sq['SQ_WAIT_INST_ANY'] / sq['SQ_WAVE_CYCLES']
    json.dump(sq, open('profiles/%s_pmc_sq_k_gram_config3.json' % tag, 'w'), indent=1, sort_keys=True)

# ---- the roofline numbers of bench.py, re-derived from the CSVs ----------------------------------------------------------------
roof = {"_note": "every number below is recomputed from profiles/%s_kernel_stats_config*.csv (rocprofv3 kernel durations), the PMC "
                 "passes and the sizes in profiles/%s_bench_config*.json; bench.py's own JSON line carries the same quantities from its "
                 "live HIP-event brackets" % (tag, tag)}
for c in cfgs:
    b, st = bench(c), stats(c)
    cfgd = b["config"]
    n, M, P = cfgd["particles_per_gpu"], cfgd["metrics"], cfgd["params"]
    K, Kp, Nn = cfgd["pred_prior_size"], cfgd["prev_pred_prior_size"], cfgd["next_set_size"]
    ent = {"ms_per_step_bench": b["ms_per_step"], "set0_ms_per_step_bench": b["set0"]["ms_per_step"] if b.get("set0") else None}
    kk = [k for k in st if k.startswith('k_kde_split')]
    kde_ms = 0.0
    if kk:
        calls, ns = st[kk[0]]
        kde_ms = ns / 1e6
        pairs = float(K) * Kp
        nch_ = 1 if P <= 16 else (2 if P <= 32 else 4)          # 16-parameter chunks of the split kernel
        # (norm steps folded into spare K-slots where the last chunk has three; else, from 4e9 pairs, tiles in the order of the norm
        # tops and one step for top and batch reference)
        mf = 6 * nch_ + (1 if P + 3 <= 16 * nch_ else (2 if pairs >= 4.0e9 else 3))
        tf = pairs * mf * 32.0 / (ns * 1e-9) / 1e12                      # matrix work ISSUED (limb products of the fp64 operands)
        alg = pairs * (3.0 * P + 1.0)                                    # K K' (3 P + 1) algorithmic flops
        atf = alg / (ns * 1e-9) / 1e12
        ent["roofline"] = {"kernel": kk[0], "bound": "mfma", "kernel_avg_ms": round(kde_ms, 4), "pairs_per_launch": pairs,
                           "flops_algorithmic": alg, "achieved_TFLOPs": round(atf, 1), "peak_TFLOPs": MFMA_PEAK_TF,
                           "frac": round(atf / MFMA_PEAK_TF, 4), "frac_algorithmic": round(atf / MFMA_PEAK_TF, 4),
                           "algorithmic_vs_fp64_vector_peak": round(atf / 78.6, 3),
                           "mfma_32x32x16_per_1024_pairs": mf, "achieved_issued_mfma_TFLOPs": round(tf, 1),
                           "mfma_issue_frac": round(tf / MFMA_PEAK_TF, 4), "traffic_hbm_bytes": traffic(c, 'k_kde_split')}
        d = G + '%s_kde_c%d' % (pp, c)
        if os.path.isdir(d):
            cn = pmc(d, kernel='k_kde_split')
            cyc = cn['GRBM_GUI_ACTIVE'] / 8
            ent["roofline"].update({
                "pmc": {k: round(v) for k, v in cn.items()}, "cycles_per_xcd": round(cyc),
                "clock_ghz": round(cyc / ns, 3),
                "mfma_busy_frac": round(cn.get('SQ_VALU_MFMA_BUSY_CYCLES', 0) / NSIMD / cyc, 3),
                # SQ_ACTIVE_INST_VALU counts quad-cycles summed over the SIMDs
                "valu_active_frac_of_kernel_cycles": round(4 * cn.get('SQ_ACTIVE_INST_VALU', 0) / NSIMD / cyc, 3),
                "valu_active_frac_of_2.4GHz_issue_peak": round(cn.get('SQ_ACTIVE_INST_VALU', 0) / (ns * 1e-9) / (NSIMD * 2.4e9 / 4), 3)})
    # the wide Gram of the ranking: every k_gram* launch except the K x P moment pass of the posterior (k_gram<1, 0, ...>)
    gk = [k for k in st if (k.startswith('k_gram') and not k.startswith('k_gram<1, 0')) or k.startswith('k_pilot_scale')]
    ngen = sum(st[k][0] for k in st if k.startswith('k_project_'))          # one projection per generation
    if gk and ngen:
        gb = 8.0 * n * (M + P)
        tot_ns = sum(st[k][1] * st[k][0] for k in gk) / ngen
        ent["roofline_hbm"] = {"kernels": gk, "launches_per_step": sum(st[k][0] for k in gk) / ngen, "algorithmic_bytes_per_step": gb,
                               "kernel_ms_per_step": round(tot_ns / 1e6, 5), "achieved_GBs": round(gb / tot_ns, 1),
                               "peak_GBs": HBM_PEAK, "frac": round(gb / tot_ns / HBM_PEAK, 4), "traffic_hbm_bytes": traffic(c, 'k_gram<3')}

    def alg(kp):
        return 8.0 * n * (M + P) + 8.0 * n * M + 8.0 * n + 16.0 * n + 16.0 * K * P + Nn * (16.0 * P + 8.0) + 8.0 * kp * P + 8.0 * (K + kp)
    sm = b["ms_per_step"] - b["roofline"]["kernel_ms"] * b["roofline"]["launches_per_step"]
    ent["roofline_streaming"] = {"algorithmic_bytes_per_step": alg(Kp), "ms": round(sm, 5), "achieved_GBs": round(alg(Kp) / sm / 1e6, 1),
                                 "frac": round(alg(Kp) / sm / 1e6 / HBM_PEAK, 4),
                                 "note": "bench step time minus bench's pair-sum kernel bracket (the profiled pass is slower: rocprofv3 overhead)"}
    if b.get("set0"):
        s0 = b["set0"]["ms_per_step"]
        ent["set0_roofline_streaming"] = {"algorithmic_bytes_per_step": alg(0), "ms": round(s0, 5),
                                          "achieved_GBs": round(alg(0) / s0 / 1e6, 1), "frac": round(alg(0) / s0 / 1e6 / HBM_PEAK, 4)}
    roof["config%d" % c] = ent
# ---- the matrix-pipe kernels of the ranking: busy cycles of the pipe against the kernel's cycles -------
MFMA_F64_PEAK_TF, MFMA_I8_PEAK_TOPS = 78.6, 5000.0       # MI355X_MICROARCH.md: fp64 matrix = fp64 vector peak; dense i8 = fp8 peak


def mfma_entry(cfg, d, st, kernel, work, peak, unit, note):
    """one kernel of one configuration: rocprofv3 duration (stats pass), the PMC pass's busy cycles, the ISSUED matrix work"""
    names = [k for k in st if k.startswith(kernel)]
    if not names or not os.path.isdir(d):
        return None
    name = max(names, key=lambda k: st[k][1])
    calls, ns = st[name]
    cn = pmc(d, kernel=kernel)
    cyc = cn.get('GRBM_GUI_ACTIVE', 0) / 8
    rate = work / (ns * 1e-9) / 1e12
    return {"kernel": name, "kernel_avg_ms": round(ns / 1e6, 4), "issued_matrix_work": work, "achieved": round(rate, 1), "peak": peak, "unit": unit,
            "frac": round(rate / peak, 4), "pmc": {k: round(v) for k, v in cn.items()}, "cycles_per_xcd_in_the_pmc_pass": round(cyc),
            "mfma_busy_frac": round(cn.get('SQ_VALU_MFMA_BUSY_CYCLES', 0) / NSIMD / cyc, 3) if cyc else None,
            "valu_active_frac": round(4 * cn.get('SQ_ACTIVE_INST_VALU', 0) / NSIMD / cyc, 3) if cyc else None, "note": note}


for c, sfx in ((4, ''), (5, ''), (5, 'f')):
    if c not in cfgs or not os.path.isdir(G + '%s_c%d%s' % (sp, c, sfx)):
        continue
    b = bench(c)
    cfgd = b["config"]
    n, M, P, A = cfgd["particles_per_gpu"], cfgd["metrics"], cfgd["params"], cfgd["pls_components"]
    stc = {short(r['Name']): (int(r['Calls']), float(r['AverageNs'])) for r in csv.DictReader(open(find(G + '%s_c%d%s' % (sp, c, sfx), 'kernel_stats.csv')))}
    if sfx:
        shutil.copy(find(G + '%s_c%d%s' % (sp, c, sfx), 'kernel_stats.csv'), 'profiles/%s_kernel_stats_config%d_fp64gram.csv' % (tag, c))
    d = G + '%s_mfma_c%d%s' % (pp, c, sfx)
    C16 = 16 * ((M + P + 15) // 16)
    nb = C16 // 16
    blocks = nb * (nb + 1) // 2 - (P // 16) * (P // 16 + 1) // 2 + P // 16        # upper-triangular 16 x 16 blocks, Y'Y only its diagonal blocks
    ents = {}
    e = mfma_entry(c, d, stc, 'k_gram_dma8', 2.0 * n * blocks * 256, MFMA_F64_PEAK_TF, "TFLOP/s",
                   "v_mfma_f64_16x16x4_f64 on the upper-triangular 16 x 16 blocks of [X|Y]'[X|Y]: 2 n 256 flops per block")
    if e: ents['k_gram_dma8'] = e
    e = mfma_entry(c, d, stc, 'k_gram_wide', 2.0 * n * blocks * 256, MFMA_F64_PEAK_TF, "TFLOP/s", "as k_gram_dma8 (ABC_GRAM_FP64 run)")
    if e: ents['k_gram_wide'] = e
    C32 = 32 * ((M + P + 31) // 32)
    t32 = (C32 // 32) * (C32 // 32 + 1) // 2
    e = mfma_entry(c, d, stc, 'k_gram_i8', 2.0 * n * t32 * 1024 * 13, MFMA_I8_PEAK_TOPS, "TOP/s",
                   "v_mfma_i32_32x32x32_i8: 13 byte-limb products (b + b' >= 2 of four balanced bytes per operand) on the upper-triangular 32 x 32 "
                   "tiles; the limb conversion (fp64 -> fixed point -> balanced bytes, per element) runs on the vector pipe beside it")
    if e: ents['k_gram_i8'] = e
    e = mfma_entry(c, d, stc, 'k_project_mfma', 2.0 * n * M * 16 * ((A + 15) // 16), MFMA_F64_PEAK_TF, "TFLOP/s",
                   "scores of all particles on v_mfma_f64_16x16x4_f64 (chained accumulator = the oracle's fma order)")
    if e: ents['k_project_mfma'] = e
    if ents:
        roof.setdefault("config%d" % c, {})["roofline_mfma" + ("_fp64gram" if sfx else "")] = ents

# ---- the Wilcoxon rule's kernels: the cascade of round 5 (the timed region's default rule: prof_c*) ---------------------------------
for c in cfgs:
    dp = G + '%s_p%d' % (sp, c)
    if os.path.isdir(dp):
        shutil.copy(find(dp, 'kernel_stats.csv'), 'profiles/%s_kernel_stats_config%d_press.csv' % (tag, c))
    b = bench(c)
    ex, cfgd = b.get("extra") or {}, b["config"]
    if cfgd.get("pls_component_rule") != "wilcoxon":
        continue
    stw = stats(c)
    if not any(k.startswith('k_wx_plan') for k in stw):
        continue
    nv, M, P, A = cfgd["particles_per_gpu"] - int(round(cfgd["particles_per_gpu"] * cfgd["train_fraction"])), cfgd["metrics"], cfgd["params"], cfgd["pls_components"]
    T = ex.get("wilcoxon_tests")
    gens = max(v[0] for k, v in stw.items() if k.startswith('k_wx_plan'))        # (one per generation; the cascade may end before k_wx_decide)
    rows, tot_ns = [], 0.0
    for k, (calls, ns) in sorted(stw.items()):
        if not k.startswith('k_wx_'):
            continue
        per_gen = calls / gens
        tot_ns += ns * calls / gens
        a, kps = None, None
        targs = [x.strip() for x in k[k.index('<') + 1:k.rindex('>')].split(',')] if '<' in k else []
        if k.startswith('k_wx_sweep') and len(targs) >= 3 and targs[2] == '0':          # level 0 (MODE = 0): every test's keys once
            a = 8.0 * nv * (A + P)
            kps = round(T * nv / (ns * 1e-9), -6) if T else None
        rows.append({"kernel": k, "launches_per_generation": per_gen, "avg_us": round(ns / 1e3, 2), "algorithmic_bytes": a,
                     "achieved_GBs": round(a / ns, 1) if a else None, "frac_of_hbm_peak": round(a / ns / HBM_PEAK, 4) if a else None, "keys_per_s": kps})
    # the scores of the validation rows come from the projection's kernels (half the rows of the projection proper): their launches with the
    # shorter duration
    total_alg = 8.0 * nv * (M + A) + 8.0 * nv * (A + P)
    roof.setdefault("config%d" % c, {})["wilcoxon_rule"] = {
        "tests": T, "validation_rows": nv, "kernels": rows, "k_wx_kernel_us_per_generation": round(tot_ns / 1e3, 1),
        "algorithmic_bytes_per_generation": total_alg,
        "rule_ms": round(b["ms_per_step"] - ex["min_press_rule_step_ms"], 4) if "min_press_rule_step_ms" in ex else None,
        "bench": {k: v for k, v in ex.items() if 'wilcoxon' in k or 'min_press' in k or k == 'ranking_pls_ms'},
        "note": "algorithmic bytes: the validation rows' metrics read once for the scores (8 nv (M + A) with the scores' write), scores and responses "
                "read once by the level-0 sweep (8 nv (A + P)); the fine levels and the exact step only touch the tests level 0 leaves undecided "
                "(how many: data; gpurun wx_debug.txt / profiles/%s_wx_debug.txt).  rule_ms = the timed step minus the argmin-PRESS step of the same run "
                "(kernels + the host's looks at the level counts)" % tag}
    with open('profiles/%s_wilcoxon_kernels_config%d.csv' % (tag, c), 'w') as f:
        f.write("kernel,launches_per_generation,avg_us,algorithmic_bytes,achieved_GBs,frac_of_hbm_peak,keys_per_s\n")
        for r in rows:
            f.write('"%s",%g,%.2f,%s,%s,%s,%s\n' % (r["kernel"], r["launches_per_generation"], r["avg_us"], r["algorithmic_bytes"], r["achieved_GBs"], r["frac_of_hbm_peak"], r["keys_per_s"]))
if os.path.exists(G + 'wx_debug.txt'):
    shutil.copy(G + 'wx_debug.txt', 'profiles/%s_wx_debug.txt' % tag)
json.dump(roof, open('profiles/%s_roofline.json' % tag, 'w'), indent=1)
for c in cfgs:
    e = roof["config%d" % c]
    print("config", c, "step %.3f ms" % e["ms_per_step_bench"], "| kde", e.get("roofline", {}).get("kernel_avg_ms"), "ms frac (algorithmic / issued)",
          e.get("roofline", {}).get("frac"), e.get("roofline", {}).get("mfma_issue_frac"), "clock", e.get("roofline", {}).get("clock_ghz"), "valu", e.get("roofline", {}).get("valu_active_frac_of_kernel_cycles"),
          "| gram", e.get("roofline_hbm", {}).get("frac"), "| streaming", e["roofline_streaming"]["frac"], "| set0",
          e.get("set0_roofline_streaming", {}).get("frac"))
