#!/usr/bin/env python3
"""HBM bytes and VALU instructions per launch of every kernel-timing slot of the library, per bench.py leg, from the
counter passes of tools/profile_round.sh.

FETCH_SIZE and WRITE_SIZE are reported in KiB; on gfx950 FETCH_SIZE counts 128-byte read requests as 64 bytes, so it is
doubled (calibrated for wide streaming reads only).  A slot (one HIP-event scope in the library) can cover several kernels:
its per-launch figure is the sum over all dispatches of its kernels divided by the dispatches of its anchor kernel (the one
launched exactly once per scope).

Usage: python3 tools/pmc_traffic.py DIR TAG   (DIR holds <leg>_{fetch,write,valu}/ rocprofv3 outputs and <leg>.meta.json)"""
import csv, glob, json, os, sys
from collections import defaultdict

# slot -> (anchor kernel, all kernels of the slot)
SLOTS = {
    # one scope = one k_pyr_down_stream (a level) or one k_pyr3_stream (levels 1 to 3)
    "k_pyr_down": (("k_pyr_down_stream", "k_pyr3_stream"), ["k_pyr_down_stream", "k_pyr3_stream", "k_pyr_down", "k_pad_border"]),
    "k_gftt_eig": ("k_gftt_eig", ["k_gftt_eig"]),
    "k_gftt_select": ("k_gftt_select", ["k_gftt_select"]),
    "k_klt": ("k_klt21", ["k_klt21"]),
    "k_compact": ("k_compact", ["k_compact"]),
    "k_append": ("k_append", ["k_append"]),
    "k_fast": ("k_fast_queue", ["k_fast_queue", "k_fast_scan_stream", "k_fast_scatter_stream"]),
    "k_orb_resize": ("k_orb_tables", ["k_orb_tables", "k_orb_resize_rows", "k_orb_resize"]),
    "k_orb_select": ("k_orb_hist", ["k_orb_hist", "k_orb_harris", "k_orb_thresh", "k_orb_emit"]),
    "k_orb_desc": ("k_orb_desc", ["k_orb_desc"]),
    "k_match": (None, ["k_match_mm", "k_match_mm_batch"]),  # one of the two per leg, one dispatch per scope
}
LEG_SLOTS = {"main": ["k_pyr_down", "k_gftt_eig", "k_gftt_select", "k_klt", "k_compact", "k_append"],
             "streams64": ["k_pyr_down", "k_gftt_eig", "k_gftt_select", "k_klt", "k_compact", "k_append"],
             "orb_match": ["k_fast", "k_orb_resize", "k_orb_select", "k_orb_desc", "k_match"],
             "reloc": ["k_match"]}


def per_kernel(root, counter):
    """kernel name -> list of counter values, one per dispatch"""
    acc = defaultdict(list)
    for f in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            if row["Counter_Name"] != counter:
                continue
            name = row["Kernel_Name"].replace("void ", "").split("(")[0].split("<")[0]
            if name.startswith("k_"):
                acc[name].append(float(row["Counter_Value"]))
    return acc


def slot_value(vals, slot, leg):
    anchor, kernels = SLOTS[slot]
    if slot == "k_match":
        kernels = ["k_match_mm_batch"] if leg == "orb_match" else ["k_match_mm"]
        anchor = kernels[0]
    anchors = anchor if isinstance(anchor, tuple) else (anchor,)
    scopes = sum(len(vals.get(a, [])) for a in anchors)
    if not scopes:
        return None, 0
    total = sum(sum(vals.get(k, [])) for k in kernels)
    return total / scopes, scopes


def main():
    root, tag = sys.argv[1], sys.argv[2]
    out = {"tag": tag, "note": "per launch of a kernel-timing slot; read = 2 x FETCH_SIZE KiB (gfx950 correction), write = WRITE_SIZE "
                               "KiB, valu_insts = SQ_INSTS_VALU (wave instructions)", "legs": {}}
    for leg, slots in LEG_SLOTS.items():
        meta_path = os.path.join(root, leg + ".meta.json")
        if not os.path.exists(meta_path):
            continue
        meta = json.load(open(meta_path))
        fetch = per_kernel(os.path.join(root, meta["pass"] + "_fetch"), "FETCH_SIZE")
        write = per_kernel(os.path.join(root, meta["pass"] + "_write"), "WRITE_SIZE")
        valu = per_kernel(os.path.join(root, meta["pass"] + "_valu"), "SQ_INSTS_VALU")
        wait = per_kernel(os.path.join(root, meta["pass"] + "_valu"), "SQ_WAIT_INST_ANY")
        wcyc = per_kernel(os.path.join(root, meta["pass"] + "_valu"), "SQ_WAVE_CYCLES")
        mfma = per_kernel(os.path.join(root, meta["pass"] + "_valu"), "SQ_VALU_MFMA_BUSY_CYCLES")
        kernels = {}
        for slot in slots:
            rd, n = slot_value(fetch, slot, leg)
            wr, _ = slot_value(write, slot, leg)
            vi, _ = slot_value(valu, slot, leg)
            wa, _ = slot_value(wait, slot, leg)
            wc, _ = slot_value(wcyc, slot, leg)
            mf, _ = slot_value(mfma, slot, leg)
            if rd is None and vi is None:
                continue
            rd_b = 2.0 * 1024.0 * rd if rd is not None else None
            wr_b = 1024.0 * wr if wr is not None else None
            kernels[slot] = {"read_bytes": rd_b, "write_bytes": wr_b, "total_bytes": (rd_b or 0.0) + (wr_b or 0.0), "valu_insts": vi,
                             "wait_inst_any_quadcycles": wa, "wave_quadcycles": wc, "mfma_busy_cycles": mf, "launches_seen": n}
        out["legs"][leg] = {"batch": meta["batch"], "command": meta["command"], "kernels": kernels}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
