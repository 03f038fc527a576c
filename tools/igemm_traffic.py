"""profiles/<tag>_pmc_{FETCH_SIZE,WRITE_SIZE,mfma}.csv (tools/summarize_prof.py pmc) -> profiles/<tag>_igemm_traffic.json:
HBM bytes per igemm launch with the gfx950 corrections of MI355X_MICROARCH.md (FETCH_SIZE in KB, x2 for wide coalesced
reads; WRITE_SIZE in KB), and the MFMA-busy fraction.   usage: python tools/igemm_traffic.py profiles r01c"""
import csv
import json
import os
import sys

d, tag = sys.argv[1], sys.argv[2]


def rows(name):
    return list(csv.DictReader(open(os.path.join(d, f"{tag}_pmc_{name}.csv"))))


def total(rs, counter, col="counter_sum"):
    return sum(float(r[col]) for r in rs if ("igemm" in r["kernel"] or "ff320" in r["kernel"]) and "fixup" not in r["kernel"] and "tile_stats" not in r["kernel"] and r["counter"] == counter)


f, w, m = rows("FETCH_SIZE"), rows("WRITE_SIZE"), rows("mfma")
launches = int(total(f, "FETCH_SIZE", "dispatches"))
dur_ns = total(f, "FETCH_SIZE", "duration_ns_sum")
fetch_kb, write_kb = total(f, "FETCH_SIZE"), total(w, "WRITE_SIZE")
hbm = (2 * fetch_kb + write_kb) * 1024
busy, gui = total(m, "SQ_VALU_MFMA_BUSY_CYCLES"), total(m, "GRBM_GUI_ACTIVE")
out = {
    "kernel": "igemm_f16_kernel (all tile / epilogue instantiations) + ff320_kernel (the fused level-0 feed-forward: the two GEMMs of 21 layers per step)",
    "launches": launches,
    "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE / --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE (separate passes, "
              "tools/profile_round.sh) on tools/profile_step.py 1 (one full-size denoise step + one 8-frame VAE chunk)",
    "fetch_size_kb_sum": fetch_kb,
    "write_size_kb_sum": write_kb,
    "correction": "bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024: FETCH_SIZE is in KB and reports half of the bytes of wide coalesced "
                  "reads on gfx950 (MI355X_MICROARCH.md, HBM); WRITE_SIZE uncalibrated",
    "hbm_bytes_per_launch": hbm / launches,
    "hbm_gbps_while_running": hbm / dur_ns,
    # SQ_VALU_MFMA_BUSY_CYCLES sums over SIMDs (4 per CU, 256 CUs); GRBM_GUI_ACTIVE sums over the 8 XCDs
    "mfma_busy_frac": busy / (gui / 8 * 256 * 4),
}
json.dump(out, open(os.path.join(d, f"{tag}_igemm_traffic.json"), "w"), indent=1)
print(json.dumps(out, indent=1))
