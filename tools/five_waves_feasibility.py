#!/usr/bin/env python3
"""Round 6, review item 4: can the moderate-ratio resample shapes run five waves per SIMD (20 per CU)?  Arithmetic on the
launch geometry of csrc/api.cpp / device.hpp (no GPU): for every way to cut a source row into equal strips, the waves a
strip's workgroup holds (a lane owns 4 source columns), the workgroup's minimal LDS (16 copies of the sRGB -> float table,
the linear -> sRGB table, the two-row ring of the fast horizontal pass: (quads + G - 1) x 48 B per row and frame slot, the
4-byte output records), how many such workgroups a CU's 160 KB holds, and the waves per CU that gives.  VGPRs: what
`python -m imageflow_amd.kernel_report 4` prints for the instantiations (five waves per SIMD allow 96).

    tools/five_waves_feasibility.py > profiles/r6_five_waves_feasibility.jsonl"""
import json

LDS = 160 * 1024
SHAPES = [  # name, in_w, out_w, fast groups G (4-column groups; the two-column form of level 1 has the same ring bytes per column), VGPRs of the instantiation
    ("cfg3-l0 3840->1600", 3840, 1600, 3, 122),
    ("cfg3-l1 1600->1200", 1600, 1200, 3, 109),
    ("cfg3-l2 1600->800", 1600, 800, 3, 122),
    ("cfg4-resize 1920->800 (planar source)", 1920, 800, 3, 120),
]


def main():
    for name, in_w, out_w, g, vgprs in SHAPES:
        quads_row = (in_w + 3) // 4
        best, busiest = None, None
        for strips in range(1, 9):
            quads = -(-quads_row // strips) + 2                      # + the window's overhang at a strip edge
            waves = -(-quads // 64)
            for frames in range(1, 9):                               # frames sharing a workgroup (and its tables)
                if waves * frames * 64 > 1024:
                    break
                n_u = -(-out_w // strips)
                lds_min = (256 << 4) * 4 + 512 + frames * 2 * ((quads + g - 1) * 48) + ((n_u * 4 + 15) & ~15) + 4096   # 16 copies, thresholds, ring, records, ~4 KB of weights
                lds_enc = lds_min + 16384                            # with the 16 KiB encode table (without it every encode is an 8-step search)
                for label, lds in (("without the encode table", lds_min), ("with the encode table", lds_enc)):
                    per_cu = min(LDS // lds, 32 // (waves * frames))
                    rec = {"shape": name, "strips_per_row": strips, "frames_per_workgroup": frames, "waves_per_workgroup": waves * frames,
                           "lds_bytes_min": lds, "tables": label, "workgroups_per_cu": per_cu, "waves_per_cu": per_cu * waves * frames,
                           "vgprs_of_the_instantiation": vgprs, "vgprs_allowed_at_that_occupancy": min(512, (512 // max(1, -(-per_cu * waves * frames // 4))) // 8 * 8)}
                    rec["busy_lanes_per_cu"] = per_cu * frames * (quads - 2)         # lanes that own source columns (a strip's last wave is partly empty)
                    rec["five_waves_per_simd"] = rec["waves_per_cu"] >= 20 and vgprs <= rec["vgprs_allowed_at_that_occupancy"]
                    print(json.dumps(rec))
                    if best is None or rec["waves_per_cu"] > best["waves_per_cu"]:
                        best = rec
                    if rec["waves_per_cu"] >= 20 and (busiest is None or rec["busy_lanes_per_cu"] > busiest["busy_lanes_per_cu"]):
                        busiest = rec
        print(json.dumps({"shape": name, "summary": "most waves per CU over all cuts", "waves_per_cu": best["waves_per_cu"], "at": {k: best[k] for k in ("strips_per_row", "frames_per_workgroup", "tables", "lds_bytes_min")},
                          "registers": f"{vgprs} VGPRs in the product's instantiation; {best['vgprs_allowed_at_that_occupancy']} allowed at that occupancy",
                          "most_busy_lanes_per_cu_among_cuts_with_20_or_more_waves": busiest and {k: busiest[k] for k in ("busy_lanes_per_cu", "waves_per_cu", "strips_per_row", "frames_per_workgroup", "tables", "vgprs_allowed_at_that_occupancy")},
                          "product_today_busy_lanes_per_cu": {"cfg3-l0 3840->1600": 960, "cfg3-l1 1600->1200": 800, "cfg3-l2 1600->800": 800, "cfg4-resize 1920->800 (planar source)": 960}[name],
                          "product_today_waves_per_cu": {"cfg3-l0 3840->1600": 15, "cfg3-l1 1600->1200": 14, "cfg3-l2 1600->800": 14, "cfg4-resize 1920->800 (planar source)": 16}[name]}))


if __name__ == "__main__":
    main()
