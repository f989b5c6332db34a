#!/usr/bin/env python3
"""Kernel A/B harness (development tool): times the ALL-mode scan of BASELINE config 2 under several tunable
settings in ONE process, interleaved rounds, and prints median/min kernel ms per variant.  The ablation variants
(tile_debug bits 1, 2, 4, 8 ... 4096) need the -DACGPU_ABLATION build: tools/build_variant.sh abl -DACGPU_ABLATION, then
ACGPU_LIB=ahocorasick_amd/lib_abl/libacgpu.so python tools/kbench.py ..."""
import argparse
import ctypes
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--units-log2", type=int, default=29)
    ap.add_argument("--rounds", type=int, default=7)
    ap.add_argument("--variants", type=str, default="")
    ap.add_argument("--config", default="C2", choices=["C2", "C2S", "C4", "C5", "C5L"])
    ap.add_argument("--set", action="store_true", help="8-byte Set records instead of 12-byte Map records")
    args = ap.parse_args()
    import torch
    from ahocorasick_amd import _native as N, synth
    from ahocorasick_amd.strings import Automaton

    n = 1 << args.units_log2
    wwl = args.config == "C5L"  # config 5's words (+ 10 % two-word keywords) and haystack through WholeWordLongestMatchMap
    if wwl:
        args.config = "C5"
    shortest = args.config == "C2S"  # config 2's dictionary and haystack through ShortestMatchMap
    if shortest:
        args.config = "C2"
    for kv in filter(None, os.environ.get("KBENCH_BUILD_TUNABLES", "").split(",")):  # builder knobs (e.g. no_tails=1) for A/B of table forms
        k, v = kv.split("=")
        N.set_tunable(k, int(v))
    kws = synth.config_keywords(args.config)
    if args.config == "C5":
        from ahocorasick_amd.unicode_tables import default_word_chars
        n = min(n, 1 << 28)
        if wwl:
            sp = np.array([32], dtype=np.uint16)
            kws = list(kws) + [np.concatenate([kws[i], sp, kws[i + 1]]) for i in range(0, 20000, 2)]
        if os.environ.get("KBENCH_EXTRA_KW"):  # one more keyword of that many units (24: the 32-unit form of k_ww_pp; 40: k_ww_tile)
            kws = list(kws) + [np.full(int(os.environ["KBENCH_EXTRA_KW"]), ord("q"), dtype=np.uint16)]
        a = Automaton(N.MODE_WWLONGEST if wwl else N.MODE_WHOLEWORD, kws, False, word_chars=default_word_chars())
        d_hay = torch.empty(n, dtype=torch.int16, device="cuda")
        synth.token_stream_on_device(d_hay.data_ptr(), n, synth.CONFIGS["C5"]["hay_seed"], synth.config_keywords("C5"), synth.swapcase_table())
        cap = n // 8
        dflt = {"wwl": {}} if wwl else {"ww": {}, "ww_noverify": {"tile_debug": 1}, "ww_nolookup": {"tile_debug": 2}, "ww_nobloom": {"tile_debug": 4}}
    else:
        a = Automaton(N.MODE_LONGEST if args.config == "C4" else (N.MODE_SHORTEST if shortest else N.MODE_ALL), kws, True)
        d_hay = torch.empty(n, dtype=torch.int16, device="cuda")
        tab = np.ascontiguousarray(synth.ALPHA_AB_75 if args.config == "C4" else synth.ALPHA_LOWER)
        N.check(N.lib().acgpu_synth_fill(d_hay.data_ptr(), n, 0, synth.CONFIGS[args.config]["hay_seed"],
                                         tab.ctypes.data_as(ctypes.c_void_p), len(tab), None), "synth")
        cap = n // 2 if args.config == "C4" else n // 128
        dflt = {"longest": {}} if args.config == "C4" else {"shortest": {}} if shortest else {
            "tile": {}, "split": {"force_kernel": 3}, "tile_noverify": {"force_kernel": 2, "tile_debug": 1},
            "tile_stream": {"force_kernel": 2, "tile_debug": 5}, "dfa": {"force_kernel": 1}, "dfa_one_chain_r3": {"force_kernel": 1, "tile_debug": 1 << 43, "lds_table_bytes": 96 * 1024}}
    torch.cuda.synchronize()
    d_out = torch.empty((cap, 3), dtype=torch.int32, device="cuda")
    variants = json.loads(args.variants) if args.variants else dflt
    defaults = {"force_kernel": 0, "tile_debug": 0, "region_units": 0, "chunk_units": 0, "lds_table_bytes": 127 * 1024,
                "blocks_per_cu": 1, "reserve_cus": 0, "longest_form": 0, "all_form": 0, "tile_form": 0}
    res = {k: [] for k in variants}
    info = {}
    for r in range(args.rounds + 1):
        for name, knobs in variants.items():
            for k, v in defaults.items():
                N.set_tunable(k, v)
            for k, v in knobs.items():
                N.set_tunable(k, v)
            nout, rc, prof, _ = a.match_device(d_hay.data_ptr(), n, not args.set, d_out.data_ptr(), cap,
                                               stream=torch.cuda.current_stream().cuda_stream, profile=True)
            if (knobs.get("tile_debug", 0) & 0xfff or (knobs.get("tile_debug", 0) >> 32) & 0x20ff) and N.set_tunable("ablation_build", 0) != 1:
                raise SystemExit("kbench: the ablation bits of tile_debug need the -DACGPU_ABLATION build "
                                 "(tools/build_variant.sh abl -DACGPU_ABLATION; ACGPU_LIB=ahocorasick_amd/lib_abl/libacgpu.so)")
            if r > 0:
                res[name].append(prof["scan_ms"])
            info[name] = (nout, rc, prof["scan_kernel"], prof["finalize_ms"])
    for name in variants:
        t = np.array(res[name])
        gb = 2 * n / 1e9
        print("%-18s median %.4f ms  min %.4f ms  (%.0f GB/s)  n_out=%d rc=%d %s fin=%.3f" % (
            name, np.median(t), t.min(), gb / (np.median(t) * 1e-3), info[name][0], info[name][1], info[name][2], info[name][3]))


if __name__ == "__main__":
    main()
