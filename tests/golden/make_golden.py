#!/usr/bin/env python3
"""Mints the golden fixtures from the UNMODIFIED reference (oracle/_ref/liblac_ref.so, built in place
from /root/reference by oracle/Makefile).  Run in the build container only:

    make -C oracle ref && python tests/golden/make_golden.py

Outputs (committed): tests/golden/small/*.lac   tiny complete .lac files for byte-for-byte diffs
                     tests/golden/digests.json  sha256 + length of the reference's .lac for seeded
                                                synthetic inputs (regenerated bit-exactly by synth.py)
Fixtures are data only: inputs are described by generator parameters, outputs are the reference's bytes.
"""
import hashlib
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import __graft_entry__ as ge  # noqa: E402
import refshim  # noqa: E402

synth = ge.load_pkg().synth

SMALL = [
    # name, frames, channels, bit_depth, rate, stereo_mode, kind, stereo, seed
    ("one_frame_st16", 1, 2, 16, 48000, 2, "noise", "wide", 1),
    ("n31_st24", 31, 2, 24, 96000, 2, "music", "wide", 2),
    ("n32_st16", 32, 2, 16, 44100, 2, "music", "narrow", 3),
    ("n33_mono16", 33, 1, 16, 48000, 0, "tone", "wide", 4),
    ("n255_st16", 255, 2, 16, 48000, 2, "walk", "wide", 5),
    ("n256_st24", 256, 2, 24, 192000, 2, "noise", "independent", 6),
    ("n257_st16_ms", 257, 2, 16, 48000, 1, "music", "wide", 7),
    ("n2400_mono16_selftest", 2400, 1, 16, 48000, 0, "tone", "wide", 8),  # BASELINE configs[0] shape
    ("n4095_st16", 4095, 2, 16, 48000, 2, "noise", "independent", 9),
    ("n4096_st16", 4096, 2, 16, 48000, 2, "mixed", "wide", 10),
    ("n4097_st16", 4097, 2, 16, 48000, 2, "noise", "independent", 11),
    ("n16421_st16", 16384 + 37, 2, 16, 48000, 2, "mixed", "wide", 12),
    ("n16421_st24_lr", 16384 + 37, 2, 24, 96000, 0, "music", "wide", 13),
    ("silence_st16", 20000, 2, 16, 48000, 2, "silence", "identical", 14),
    ("sparse_mono24", 9000, 1, 24, 48000, 0, "sparse", "wide", 15),
]

DIGESTS = [
    # name, frames, channels, bit_depth, rate, stereo_mode, kind, stereo, seed, cpu_test, gpu_test
    ("cfg2_10min_st16_48k_auto", 28_800_000, 2, 16, 48000, 2, "music", "wide", 2026, False, True),
    ("cfg2_60s_st16_48k_auto", 2_880_000, 2, 16, 48000, 2, "music", "wide", 2026, True, True),
    ("cfg3_60s_st24_96k_mixed", 5_760_000, 2, 24, 96000, 2, "mixed", "wide", 7, True, True),
    ("noise_20s_st16_48k", 960_000, 2, 16, 48000, 2, "noise", "independent", 3, True, True),
    ("mono_30s_16_44k", 1_323_000, 1, 16, 44100, 0, "mixed", "wide", 5, True, True),
    ("forced_ms_20s_24_192k", 3_840_000, 2, 24, 192000, 1, "music", "narrow", 9, False, True),
    ("forced_lr_20s_16_96k", 1_920_000, 2, 16, 96000, 0, "mixed", "wide", 11, False, True),
    # BASELINE configs[3]: 2 h stereo 16/48 (21 094 blocks) split in 8 contiguous block ranges; this is the last
    # range [18457, 21094) -- 2637 blocks incl. the 12 288-frame final block -- encoded as a stream of its own
    # (blocks are independent, so payload and table equal that range of the whole stream's).
    ("cfg4_2h_shard8of8_st16_48k", 345_600_000 - 18457 * 16384, 2, 16, 48000, 2, "music", "wide", 2026, False, True,
     18457 * 16384),
]
# BASELINE configs[2] at its stated size (10 min stereo 24-bit 96 kHz, 57.6 M frames, 3516 blocks), one and ten minutes of white
# noise (every block "uncertain": all twelve probes, about five exactly costed candidates per slot), and the sixteen
# streams of BASELINE configs[4] ({mono, stereo} x {16, 24 bit} x {44.1, 48, 96, 192 kHz}, 60 s each; stream i: seed
# 500 + i, "mixed" material for odd i, "music" for even i): bench.py times them after the headline loop and checks every
# .lac against these digests ("other_workloads").
DIGESTS.append(("cfg3_10min_st24_96k_mixed", 57_600_000, 2, 24, 96000, 2, "mixed", "wide", 7, False, True))
DIGESTS.append(("noise_60s_st16_48k", 2_880_000, 2, 16, 48000, 2, "noise", "independent", 3, False, True))
DIGESTS.append(("noise_10min_st16_48k", 28_800_000, 2, 16, 48000, 2, "noise", "independent", 3, False, False))
CFG5 = []
for _ch in (1, 2):
    for _bd in (16, 24):
        for _sr in (44100, 48000, 96000, 192000):
            _i = len(CFG5)
            CFG5.append((f"cfg5_{_i:02d}_{'st' if _ch == 2 else 'mono'}{_bd}_{_sr}", 60 * _sr, _ch, _bd, _sr, 2 if _ch == 2 else 0,
                         "mixed" if _i & 1 else "music", "wide", 500 + _i, False, True))
DIGESTS.extend(CFG5)
# The other seven eighths of the same 2 h stream (block ranges [r*B/8, (r+1)*B/8), B = 21 094): bench.py --gpus 2/4/8
# has every rank check each eighth of its shard against these (the N = 2 and N = 4 boundaries are N = 8 boundaries).
CFG4_FRAMES, CFG4_BLOCKS = 345_600_000, 21094
for _r in range(7):
    _b0, _b1 = _r * CFG4_BLOCKS // 8, (_r + 1) * CFG4_BLOCKS // 8
    DIGESTS.append((f"cfg4_2h_shard{_r + 1}of8_st16_48k", (_b1 - _b0) * 16384, 2, 16, 48000, 2, "music", "wide", 2026,
                    False, _r in (0, 3), _b0 * 16384))


def main():
    if not refshim.available():
        raise SystemExit("oracle/_ref/liblac_ref.so missing: run `make -C oracle ref` where /root/reference exists")
    os.makedirs(os.path.join(HERE, "small"), exist_ok=True)
    index = []
    for name, frames, ch, bd, sr, sm, kind, st, seed in (SMALL if not (len(sys.argv) > 2 and sys.argv[1] == "--only") else []):
        left, right = synth.synth_pcm(frames, ch, bd, sr, seed=seed, kind=kind, stereo=st)
        data = refshim.encode(left, right, sr, bd, sm)
        with open(os.path.join(HERE, "small", name + ".lac"), "wb") as f:
            f.write(data)
        index.append(dict(name=name, stereo_mode=sm, lac_bytes=len(data), lac_sha256=hashlib.sha256(data).hexdigest(),
                          gen=dict(frames=frames, channels=ch, bit_depth=bd, sample_rate=sr, seed=seed, kind=kind,
                                   stereo=st)))
        print(name, len(data))
    if index:
        with open(os.path.join(HERE, "small", "index.json"), "w") as f:
            json.dump(index, f, indent=1)
    out = []
    only = sys.argv[2] if len(sys.argv) > 2 and sys.argv[1] == "--only" else None
    if only:  # re-mint the entries whose name starts with the prefix, keep the others as they are
        with open(os.path.join(HERE, "digests.json")) as f:
            out = [d for d in json.load(f) if not d["name"].startswith(only)]
    for ent in DIGESTS:
        name, frames, ch, bd, sr, sm, kind, st, seed, cpu_test, gpu_test = ent[:11]
        if only and not name.startswith(only):
            continue
        start = ent[11] if len(ent) > 11 else 0
        left, right = synth.synth_pcm(frames, ch, bd, sr, seed=seed, kind=kind, stereo=st, start=start)
        data = refshim.encode(left, right, sr, bd, sm)
        out.append(dict(name=name, stereo_mode=sm, lac_bytes=len(data), lac_sha256=hashlib.sha256(data).hexdigest(),
                        cpu_test=cpu_test, gpu_test=gpu_test,
                        gen=dict(frames=frames, channels=ch, bit_depth=bd, sample_rate=sr, seed=seed, kind=kind,
                                 stereo=st, start=start)))
        print(name, len(data), out[-1]["lac_sha256"][:16])
    order = {ent[0]: i for i, ent in enumerate(DIGESTS)}
    out.sort(key=lambda d: order.get(d["name"], len(order)))
    with open(os.path.join(HERE, "digests.json"), "w") as f:
        json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()
