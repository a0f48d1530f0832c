"""profiles/traffic.json is WRITTEN, not edited.

  entry <workload> <profdir> [--commit C]   (on the GPU box, last step of tools/gpu_profile.sh)
      reads the FETCH_SIZE / WRITE_SIZE passes of <profdir> (rocprofv3 --pmc, separate passes), takes the per-launch averages of
      the workload's dominant kernel and writes <profdir>/traffic_entry.json: bytes per launch = 1024 * (2 * FETCH_SIZE +
      WRITE_SIZE) -- the counters are in KiB and gfx950 reports half of a wide coalesced read (MI355X_MICROARCH.md; confirmed on
      C1: 16 408 KiB reported for a 33.5 MB input) --, the kernel's name, the commit the caller names, the held clock if the
      trace log has one, and `sources_sha256`: a hash over the kernel's source files AS THEY WERE MEASURED.
  merge <traffic_entry.json> ...            (in the build container, after gpurun merged gpurun_out/)
      puts the entries into profiles/traffic.json.

bench.py recomputes the hash of the same files when it reports `roofline.traffic`: an entry measured on other sources is refused
(traffic null, traffic_source says why) instead of being carried along by hand from round to round."""
from __future__ import annotations

import csv
import glob
import hashlib
import json
import sys
import time
from collections import defaultdict
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
CSRC = "conch_amd/csrc/"
# workload -> (substring of the dominant kernel's name, algorithmic bytes, the sources that define that kernel)
WORKLOADS = {
    "c1": ("quant_flat_kernel", 4096 * 4096 * 3, [CSRC + "quant.hip", CSRC + "quant_common.hpp"]),
    "c2": ("skinny_splitk_kernel", 128 * 4096 + 4096 * 4096 + 2 * 128 * 4096 + 4 * (128 + 4096), [CSRC + "gemm_skinny.hip"]),
    "c3": ("conch_gemm1w_fp8_bf16", 4096 * 4096 + 4096 * 11008 + 2 * 4096 * 11008 + 4 * (4096 + 11008), [CSRC + "asm/gen_gemm1w.py", CSRC + "gemm_asm.hip"]),
    "c4": ("mixed_strip_kernel", 2 * 1024 * 4096 + 4096 * 11008 // 2 + 2 * 32 * 11008 + 2 * 1024 * 11008, [CSRC + "gemm_mixed_strip.hip", CSRC + "mixed_dequant.hpp"]),
    "c4readme": ("mixed_gemm_kernel", 2 * 4096 * 8192 + 8192 * 4096 // 2 + 2 * 64 * 4096 + 2 * 4096 * 4096, [CSRC + "gemm_mixed.hip", CSRC + "mixed_dequant.hpp"]),
}


def sources_sha256(workload: str, root: Path = ROOT) -> str:
    h = hashlib.sha256()
    for rel in WORKLOADS[workload][2]:
        h.update(rel.encode())
        h.update((root / rel).read_bytes())
    return h.hexdigest()[:16]


def counter_avg(profdir: Path, sub: str, kernel: str, counter: str) -> tuple[float, int]:
    vals = []
    for f in glob.glob(f"{profdir}/{sub}/**/*counter_collection.csv", recursive=True):
        per_dispatch: dict = defaultdict(float)
        for row in csv.DictReader(open(f)):
            if kernel in row["Kernel_Name"] and row["Counter_Name"] == counter:
                per_dispatch[row.get("Dispatch_Id", len(per_dispatch))] += float(row["Counter_Value"])
        vals += list(per_dispatch.values())
    if not vals:
        raise SystemExit(f"no {counter} samples of a kernel named *{kernel}* under {profdir}/{sub}")
    return sum(vals) / len(vals), len(vals)


def entry(workload: str, profdir: Path, commit: str) -> dict:
    kernel, alg, sources = WORKLOADS[workload]
    fetch, nf = counter_avg(profdir, "pmc_fetch", kernel, "FETCH_SIZE")
    write, nw = counter_avg(profdir, "pmc_write", kernel, "WRITE_SIZE")
    hit, _ = counter_avg(profdir, "pmc_write", kernel, "TCC_HIT_sum")
    miss, _ = counter_avg(profdir, "pmc_write", kernel, "TCC_MISS_sum")
    return {
        "hbm_bytes_per_launch": int(round(1024 * (2 * fetch + write))), "fetch_kib": round(fetch, 1), "write_kib": round(write, 1),
        "tcc_hit_rate": round(hit / max(hit + miss, 1.0), 4), "launches_sampled": min(nf, nw), "algorithmic_bytes": alg,
        "ratio_to_algorithmic": round(1024 * (2 * fetch + write) / alg, 3), "kernel": kernel, "commit": commit,
        "measured": time.strftime("%Y-%m-%d %H:%M:%S UTC", time.gmtime()), "sources": sources, "sources_sha256": sources_sha256(workload),
        "source": f"written by tools/write_traffic.py from the rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of tools/gpu_profile.sh ({profdir.name}, commit {commit}); "
                  "FETCH_SIZE doubled per the guide's gfx950 correction, counters in KiB",
    }


def main() -> None:
    if len(sys.argv) >= 4 and sys.argv[1] == "entry":
        workload, profdir = sys.argv[2], Path(sys.argv[3])
        commit = sys.argv[sys.argv.index("--commit") + 1] if "--commit" in sys.argv else "unknown"
        rec = entry(workload, profdir, commit)
        (profdir / "traffic_entry.json").write_text(json.dumps({workload: rec}, indent=2) + "\n")
        print(json.dumps({workload: rec}, indent=2))
    elif len(sys.argv) >= 3 and sys.argv[1] == "merge":
        f = ROOT / "profiles" / "traffic.json"
        table = json.loads(f.read_text()) if f.exists() else {}
        table["_comment"] = ("Written by tools/write_traffic.py (never edited by hand): L2 <-> fabric bytes per launch of each workload's dominant kernel from "
                             "rocprofv3 --pmc (separate FETCH_SIZE / WRITE_SIZE passes).  These counters sit on the L2's memory side, so Infinity-Cache hits are "
                             "included.  bench.py refuses an entry whose sources_sha256 differs from the sources it runs.")
        for path in sys.argv[2:]:
            table.update(json.loads(Path(path).read_text()))
        f.write_text(json.dumps(table, indent=2) + "\n")
        print(f"merged {len(sys.argv) - 2} entr(y/ies) into {f}")
    else:
        raise SystemExit(__doc__)


if __name__ == "__main__":
    main()
