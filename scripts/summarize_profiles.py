"""Copies the rocprofv3 summaries a gpurun call left under gpurun_out/ into profiles/ (tracked) and derives the
per-kernel HBM traffic from the separate FETCH_SIZE / WRITE_SIZE PMC passes, with the gfx950 correction of
/opt/skills/guides/MI355X_MICROARCH.md (FETCH_SIZE reports 1/2 of a wide coalesced read; WRITE_SIZE exact).
usage: python scripts/summarize_profiles.py <round-tag> <stats-prefix> <pmc-prefix>"""
import csv, glob, json, re, shutil, sys

tag, sp, pp = sys.argv[1], sys.argv[2], sys.argv[3]


def find(d, suffix):
    return sorted(glob.glob(d + '/**/*' + suffix, recursive=True))[0]


def short(k):
    m = re.search(r'(k_[a-z0-9_]+(<[^>]*>)?)', k)
    return m.group(1) if m else k.split('(')[0].strip()


def pmc(d, counter):
    out = {}
    for r in csv.DictReader(open(find(d, 'counter_collection.csv'))):
        if r['Counter_Name'] == counter:
            out.setdefault(r['Kernel_Name'], []).append(float(r['Counter_Value']))
    return {k: sum(v) / len(v) for k, v in out.items()}


res = {"_note": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in SEPARATE passes of `python3 bench.py --config N --steps 2 "
                "--warmup 1 --no-cpu-baseline`; per-dispatch averages in KiB; hbm_bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024 "
                "(gfx950: FETCH_SIZE counts 64 B per 128-B request).", "configs": {}}
rows = []
for cfg in (2, 3):
    shutil.copy(find('gpurun_out/%s_c%d' % (sp, cfg), 'kernel_stats.csv'),
                'profiles/%s_kernel_stats_config%d.csv' % (tag, cfg))
    fe, wr = pmc('gpurun_out/%s_fetch_c%d' % (pp, cfg), 'FETCH_SIZE'), pmc('gpurun_out/%s_write_c%d' % (pp, cfg), 'WRITE_SIZE')
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
print("k_gram traffic:", {c: [v["hbm_bytes_per_launch"] for k, v in e.items() if k.startswith(("k_gram<3", "k_gram_dma<3"))] for c, e in res["configs"].items()})

# SQ counters of the Gram kernel (two passes: pmcf_sq_c3, pmcf_sq2_c3)
sq = {}
for d in ('gpurun_out/%s_sq_c3' % pp, 'gpurun_out/%s_sq2_c3' % pp):
    try:
        fs = [find(d, 'counter_collection.csv')]
    except IndexError:
        continue
    acc = {}
    for r in csv.DictReader(open(fs[0])):
        if 'k_gram_dma<3' in r['Kernel_Name'] or 'k_gram<3' in r['Kernel_Name']:
            acc.setdefault(r['Counter_Name'], []).append(float(r['Counter_Value']))
            sq['_kernel'] = short(r['Kernel_Name'])
    for k, v in acc.items():
        sq[k] = sum(v) / len(v)
if sq:
    sq['_note'] = ("rocprofv3 --pmc (two passes) on `python3 bench.py --config 3 --steps 2 --warmup 1 --no-cpu-baseline`, per-dispatch "
                   "averages; MfmaUtil = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE/8 * 1024 SIMDs)")
    if 'GRBM_GUI_ACTIVE' in sq and 'SQ_VALU_MFMA_BUSY_CYCLES' in sq:
        sq['MfmaUtil_percent'] = 100 * sq['SQ_VALU_MFMA_BUSY_CYCLES'] / (sq['GRBM_GUI_ACTIVE'] / 8 * 1024)
    if 'SQ_WAIT_INST_ANY' in sq and 'SQ_WAVE_CYCLES' in sq:
        sq['wait_fraction_of_wave_cycles'] = sq['SQ_WAIT_INST_ANY'] / sq['SQ_WAVE_CYCLES']
    json.dump(sq, open('profiles/%s_pmc_sq_k_gram_config3.json' % tag, 'w'), indent=1, sort_keys=True)
    print("sq:", sq)

# the weight kernel (k_kde_split / k_kde): clock and matrix-pipe utilisation from the pmcf_kde_c3 pass + the kernel stats
try:
    kd = {}
    for r in csv.DictReader(open(find('gpurun_out/%s_kde_c3' % pp, 'counter_collection.csv'))):
        if 'k_kde' in r['Kernel_Name']:
            kd.setdefault(short(r['Kernel_Name']), {}).setdefault(r['Counter_Name'], []).append(float(r['Counter_Value']))
    dur = {}
    for r in csv.DictReader(open(find('gpurun_out/%s_c3' % sp, 'kernel_stats.csv'))):
        if 'k_kde' in r['Name']:
            dur[short(r['Name'])] = float(r['AverageNs'])
    out = {"_note": "rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU on `python3 bench.py "
                    "--config 3 --steps 2 --warmup 1 --no-cpu-baseline` (per-dispatch averages) + the kernel duration of the "
                    "--kernel-trace --stats pass; GRBM_GUI_ACTIVE is summed over the 8 XCDs: clock = cycles / 8 / duration; "
                    "mfma_busy_frac = SQ_VALU_MFMA_BUSY_CYCLES / 1024 SIMDs / cycles per XCD"}
    for k, c in kd.items():
        c = {n: sum(v) / len(v) for n, v in c.items()}
        if c.get('GRBM_GUI_ACTIVE', 0) < 1e6:          # the kernel that was not its turn returns at once
            continue
        cyc = c['GRBM_GUI_ACTIVE'] / 8
        ent = {"counters": {n: round(v) for n, v in c.items()}, "cycles_per_xcd": round(cyc),
               "mfma_busy_frac": round(c.get('SQ_VALU_MFMA_BUSY_CYCLES', 0) / 1024 / cyc, 3)}
        if k in dur:
            ent["kernel_avg_ms"] = round(dur[k] / 1e6, 4)
            ent["clock_ghz"] = round(cyc / dur[k], 3)
        out[k] = ent
    json.dump(out, open('profiles/%s_pmc_kde_config3.json' % tag, 'w'), indent=1, sort_keys=True)
    print("kde:", {k: v for k, v in out.items() if k != "_note"})
except (IndexError, FileNotFoundError) as e:
    print("no kde pmc pass:", e)
