#!/usr/bin/env python3
"""Generates profiles/<tag>_summary.md - every figure the documents quote about a round's profiles - from the JSON / CSV files themselves
(profiles/README.md and DESIGN.md quoted per-layer figures of an earlier collection than the committed JSON).
tests/test_docs_refs.py regenerates the text and fails when the committed summary differs, and when DESIGN.md's block between the
`<!-- generated: profiles/<tag>_summary.md -->` markers is not that text.

    python tools/summarize_profiles.py r04            # writes profiles/r04_summary.md
    python tools/summarize_profiles.py r04 --stdout   # prints it (what the test compares)
"""
import csv
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = os.path.join(ROOT, "profiles")
PEAK_TF = 2500.0


def _json(name):
    path = os.path.join(P, name)
    if not os.path.exists(path):
        return None
    with open(path) as f:
        return json.load(f)


def _stats(name):
    """{short kernel name: (calls, total ms)} and the k_adam call count of a stamped rocprofv3 kernel_stats csv"""
    path = os.path.join(P, name)
    if not os.path.exists(path):
        return None, 0, ""
    rows = [ln for ln in open(path) if ln.strip()]
    stamp = rows[0].strip() if rows[0].startswith("#") else ""
    out = {}
    for r in csv.DictReader(rows[1:] if stamp else rows):
        nm = re.sub(r"\(anonymous namespace\)::", "", r["Name"]).replace("void ", "").split("(")[0]
        c, t = out.get(nm, (0, 0.0))
        out[nm] = (c + int(r["Calls"]), t + float(r["TotalDurationNs"]) / 1e6)
    steps = sum(c for n, (c, _) in out.items() if n == "k_adam")
    return out, steps, stamp


def _family(out, steps, prefixes):
    c = sum(v[0] for n, v in out.items() if n.startswith(prefixes))
    t = sum(v[1] for n, v in out.items() if n.startswith(prefixes))
    return c / max(steps, 1), t / max(steps, 1)


def summary(tag):
    L = []
    w = L.append
    w("# %s - figures generated from the files in this directory by `python tools/summarize_profiles.py %s` (do not edit)" % (tag, tag))
    w("")
    pl = _json("%s_per_layer.json" % tag)
    if pl:
        w("## Per layer (`%s_per_layer.json`; exclusive = one stream, HIP events; kernel_source_hash `%s`, commit `%s`)" % (
            tag, pl.get("kernel_source_hash"), pl.get("git_head")))
        w("")
        w("| layer | Cin -> Cout @ level | fwd ms (frac of 2.5 PF) | dgrad ms (frac) | wgrad ms (frac) |")
        w("|---|---|---|---|---|")
        by = {}
        for r in pl["rows"]:
            by.setdefault(r["layer"], {"lvl": r["level"], "cin": r["cin"], "cout": r["cout"], "par": r["parity_form"]})[r["pass"]] = (r["ms"], r["mfma_frac"])
        for name, v in by.items():
            cell = lambda k: "%.3f (%.2f)" % v[k] if k in v else "-"
            w("| %s%s | %d -> %d @ %d | %s | %s | %s |" % (name, " (parity form)" if v["par"] else "", v["cin"], v["cout"], v["lvl"], cell("fwd"), cell("dgrad"), cell("wgrad")))
        t = pl["total"]
        w("")
        w("Sum over the table: **%.2f ms**, %.0f GFLOP algorithmic = %.3f of the MFMA peak; algorithmic bytes %.0f MB = %.3f of 8 TB/s." % (
            t["ms"], t["gflop"], t["mfma_frac"], t["algorithmic_mb"], t["hbm_frac"]))
        w("")
    for kind, fname in (("timed region, two streams (durations of concurrent kernels overlap)", "%s_kernel_stats.csv" % tag),
                        ("one stream (exclusive durations)", "%s_kernel_stats_one_stream.csv" % tag)):
        out, steps, stamp = _stats(fname)
        if not out:
            continue
        w("## rocprofv3 kernel stats, %s (`%s`, %d profiled steps; %s)" % (kind, fname, steps, stamp.lstrip("# ")))
        w("")
        fc, ft = _family(out, steps, ("k_conv_fwd_ws", "k_conv_fwd_mfma"))
        wc, wt = _family(out, steps, ("k_conv_wgrad_mfma", "k_conv_wgrad_kd", "k_conv_wgrad_up_kd"))
        tot = sum(v[1] for v in out.values()) / max(steps, 1)
        w("* forward / input-gradient family (`k_conv_fwd_*`): %.0f launches per step, **%.2f ms per step**, %.0f us per launch" % (fc, ft, ft / max(fc, 1) * 1e3))
        w("* weight-gradient family (`k_conv_wgrad_mfma`, `k_conv_wgrad_kd`, `k_conv_wgrad_up_kd`): %.0f launches per step, **%.2f ms per step**, %.0f us per launch" % (wc, wt, wt / max(wc, 1) * 1e3))
        w("* everything else: %.2f ms per step; all kernels: %.2f ms per step" % (tot - ft - wt, tot))
        rest = sorted(((v[1] / max(steps, 1), n) for n, v in out.items() if not n.startswith(("k_conv_fwd_ws", "k_conv_fwd_mfma", "k_conv_wgrad_mfma", "k_conv_wgrad_kd", "k_conv_wgrad_up_kd"))), reverse=True)[:8]
        w("* largest of the rest (ms per step): " + ", ".join("`%s` %.3f" % (n[:40], t) for t, n in rest))
        w("")
    tr = _json("%s_pmc_traffic_per_step.json" % tag)
    if tr:
        fam = {"forward / input gradient (`k_conv_fwd_*`)": ("k_conv_fwd_",), "weight gradient (`k_conv_wgrad_*`, `k_wgrad_reduce`, `k_expand_up_wgrad`)": ("k_conv_wgrad", "k_wgrad_reduce", "k_expand_up_wgrad"),
               "weight repack (`k_pack_*`)": ("k_pack_",)}
        w("## HBM traffic per step from the counters (`%s_pmc_traffic_per_step.json`: FETCH_SIZE x 2 + WRITE_SIZE, separate passes; hash `%s`)" % (
            tag, tr["_meta"].get("kernel_source_hash")))
        w("")
        tot = 0.0
        for label, pre in fam.items():
            mb = sum(v["hbm_mb_corrected"] for k, v in tr.items() if k != "_meta" and k.startswith(pre))
            w("* %s: **%.2f GB**" % (label, mb / 1e3))
        tot = sum(v["hbm_mb_corrected"] for k, v in tr.items() if k != "_meta")
        w("* all kernels of a step: **%.1f GB** (algorithmic: 3x3x3 convs 17.35 GB, SURVEY 8d)" % (tot / 1e3))
        w("")
    mf = _json("%s_pmc_mfma.json" % tag)
    if mf:
        w("## Matrix-pipe utilisation from the counters (`%s_pmc_mfma.json`: %s; exclusive launches)" % (tag, mf["_meta"].get("definition")))
        w("")
        for k, v in mf.get("families", {}).items():
            w("* %s: **%.3f** (%.2f TMAC per step issued)" % (k, v["mfma_util"], v.get("tmac_per_step_if_32x32x16", 0.0)))
        ks = sorted(((v["mfma_util"], k) for k, v in mf["kernels"].items() if k.startswith("k_conv_")), reverse=True)
        w("* per kernel: " + ", ".join("`%s` %.2f" % (k[:44], u) for u, k in ks))
        w("")
    for fname, title in (("%s_bench_default.json" % tag, "default `python bench.py`"),):
        b = _json(fname)
        if not b:
            continue
        w("## %s (`%s`)" % (title, fname))
        w("")
        w("* **%.1f patches/s, %.2f ms per batch-4 step** (%s); train Dice of the last timed step %.3f" % (b["value"], b["ms_per_step"], b["data"], b["train_dice_last_step"]))
        r, rx = b.get("roofline", {}), b.get("roofline_exclusive", {})
        w("* `roofline.frac` %.3f (two-stream timed region, %d sampled steps, %.0f us per launch), `roofline_exclusive.frac` %.3f, `step_mfma_frac` %.3f, "
          "`hbm_roofline_frac_conv_algorithmic` %.3f" % (r.get("frac", 0), r.get("sampled_steps", 0), r.get("avg_launch_ms", 0) * 1e3, rx.get("frac", 0),
                                                         b.get("step_mfma_frac", 0), b.get("hbm_roofline_frac_conv_algorithmic", 0)))
        if r.get("traffic"):
            w("* `roofline.traffic` %.0f MB per launch (algorithmic %.0f MB), `roofline.mfma_util_pmc` %.3f" % (
                r["traffic"] / 1e6, (11.01e9 / max(r.get("launches_per_step", 32), 1)) / 1e6, (r.get("mfma_util_pmc") or {}).get("mfma_util", 0)))
        c = b.get("continuity")
        if c:
            w("* `continuity` (rounds 1-3 recipe, same process): %.1f patches/s, train Dice %.4f" % (c["value"], c["train_dice_last_step"]))
        v = b.get("val_dice")
        if v:
            w("* `val_dice`: soft %.4f, hard over the reconstructed 160x256x256 volume %.4f, %d steps; %.1f patches/s on that task" % (
                v["soft"], v["hard_cfg5_volume"], v["steps"], v["patches_per_s_on_this_task"]))
        ra = b.get("reference_api")
        if ra:
            w("* `reference_api` (fit_generator): host float64 generator %.1f, device batches %.1f, bare engine loop %.1f patches/s" % (
                ra["host_float64_generator_patches_per_s"], ra["device_batches_patches_per_s"], ra["resident_batch_patches_per_s"]))
            if "device_generator_default_augmentation_patches_per_s" in ra:
                w("* `reference_api`, device generator with the reference's default augmentation: %.1f patches/s = %.3f of the same generator's batches "
                  "cycled from HBM (%.1f before, %.1f after)" % (
                      ra["device_generator_default_augmentation_patches_per_s"], ra["device_generator_vs_its_ready_batches"],
                      ra["device_generator_ready_batches_patches_per_s"][0], ra["device_generator_ready_batches_patches_per_s"][1]))
        s = b.get("secondary")
        if s:
            w("* `secondary`: configs[3] %.0f slices/s (%.2f ms, %.3f of the MFMA peak); configs[4] %.1f ms per volume end to end, %.1f ms device tile loop "
              "(%.3f of the MFMA peak), host share %.1f %%" % (s["cfg3"]["slices_per_s"], s["cfg3"]["ms_per_step"], s["cfg3"]["mfma_frac"],
                                                              s["cfg4"]["end_to_end_s_per_volume"] * 1e3, s["cfg4"]["device_tile_loop_s"] * 1e3,
                                                              s["cfg4"]["mfma_frac_device"], s["cfg4"]["host_share"] * 100))
        cb = b.get("cpu_baseline")
        if cb:
            w("* `cpu_baseline`: %.3f patches/s on %d threads (%s)" % (cb["value"], cb["cores"], cb["kind"]))
        w("")
    return "\n".join(L) + "\n"


BEGIN, END = "<!-- generated: profiles/%s_summary.md -->", "<!-- end generated -->"


def design_block(tag):
    """the text between DESIGN.md's markers for `tag` (None when the markers are absent)"""
    text = open(os.path.join(ROOT, "DESIGN.md")).read()
    b = BEGIN % tag
    if b not in text:
        return None
    a = text.index(b) + len(b)
    return text[a:text.index(END, a)].strip("\n") + "\n"


def demote(text):
    """the summary as a block of DESIGN.md: its headings one level down, without the title line"""
    lines = text.splitlines()[1:]
    return "\n".join(("##" + ln) if ln.startswith("## ") else ln for ln in lines).strip("\n") + "\n"


if __name__ == "__main__":
    tag = sys.argv[1] if len(sys.argv) > 1 else "r04"
    text = summary(tag)
    if "--stdout" in sys.argv:
        sys.stdout.write(text)
    else:
        with open(os.path.join(P, "%s_summary.md" % tag), "w") as f:
            f.write(text)
        print("wrote profiles/%s_summary.md (%d lines)" % (tag, text.count("\n")))
        dpath = os.path.join(ROOT, "DESIGN.md")
        d = open(dpath).read()
        b = BEGIN % tag
        if b in d:
            a = d.index(b) + len(b)
            e = d.index(END, a)
            with open(dpath, "w") as f:
                f.write(d[:a] + "\n" + demote(text) + d[e:])
            print("updated the generated block of DESIGN.md")
