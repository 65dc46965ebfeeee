#!/usr/bin/env python3
"""Classify ThreadSanitizer reports of a run through libpdhg_hip_tsan.so: a report counts against the library only when
one of its two racing ACCESSES (the first frame outside the TSan runtime) lies in libpdhg_hip_tsan.so; reports whose
accesses are both inside the uninstrumented HIP / HSA runtime (its allocator, its queues) are noise of the method.
usage: tsan_classify.py gpurun_out/tsan_report.*"""
import re
import sys

total, kinds, ours = 0, {}, []
for path in sys.argv[1:]:
    for r in open(path, errors="replace").read().split("=================="):
        if "WARNING: ThreadSanitizer" not in r:
            continue
        total += 1
        blocks = [b for b in re.split(r"\n\s*\n", r) if "of size" in b[:120] or "Previous" in b[:40]]
        tops = []
        for b in blocks[:2]:
            frames = [f for f in re.findall(r"#\d+ (.*)", b) if "libclang_rt" not in f]
            tops.append(frames[0] if frames else "?")
        key = tuple(sorted("library" if "libpdhg_hip_tsan" in t else ("hip-runtime" if ("libamdhip64" in t or "libhsa" in t) else "other")
                           for t in tops))
        kinds[key] = kinds.get(key, 0) + 1
        if "library" in key:
            ours.append(r.strip()[:3000])
print(f"{total} ThreadSanitizer reports; racing accesses by module: " + ", ".join(f"{' / '.join(k)}: {v}" for k, v in sorted(kinds.items())))
print(f"reports with a racing access inside libpdhg_hip_tsan.so: {len(ours)}")
for r in ours[:5]:
    print("-----\n" + r)
