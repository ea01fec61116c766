#!/usr/bin/env python3
"""Print the figures of one bench.py line that a round's notes quote: python tools/bench_digest.py FILE"""
import json
import sys


def sect(tag, d):
    r, u = d.get("roofline") or {}, d.get("update_roofline") or {}
    print("%-16s %9.1f samples/s  %7.3f ms/round  enc %7.3f  upd %6.3f  | dominant %s frac %s | update %.4f ms/step %s %.3f (hbm %.3f mfma %s)"
          % (tag, d["value"], d["ms_per_step"], d.get("t_encode_ms", 0), d.get("t_update_ms", 0), r.get("kernel"), r.get("frac"),
             u.get("ms_per_step", 0), u.get("bound"), u.get("frac", 0) or 0, u.get("hbm_frac", 0), u.get("mfma_frac")))


def main(path):
    d = json.loads(open(path).read().strip().splitlines()[-1])
    sect("headline n=%d" % d["n_gpus"], d)
    for k in ("c3", "c2", "c2_direct_conv", "c2_stem_bf16x3", "c2_latent_cache", "c3_latent_cache"):
        if k in d:
            sect(k, d[k])
    if d.get("c3", {}).get("encoder_fwd_hbm_frac") is not None:
        print("c3 encoder_fwd_hbm_frac %.4f (target %s), encoder_tflops %s" % (d["c3"]["encoder_fwd_hbm_frac"], d["c3"].get("encoder_fwd_hbm_frac_target"), d["c3"].get("encoder_tflops")))
    for k, v in list((d.get("c3", {}).get("roofline") or {}).get("per_kernel", {}).items())[:12]:
        print("   c3 ", k, v)
    if "cpu_baseline" in d:
        print("cpu_baseline", d["cpu_baseline"]["value"], d["cpu_baseline"]["cores"])


if __name__ == "__main__":
    main(sys.argv[1])
