#!/usr/bin/env python3
"""Run rocprofv3 passes (kernel trace + PMC groups, each in its own run) around a command and
print one compact table per kernel.  Runs on the GPU box:  python tools/prof.py --out DIR -- cmd...
FETCH_SIZE is printed raw AND doubled (gfx950 reports 1/2 of wide coalesced reads; calibrated on
k_convert, whose byte count is known -- MI355X_MICROARCH.md HBM section)."""
import argparse
import glob
import os
import sqlite3
import subprocess
import sys

GROUPS = {
    "sq": "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_RD",
    "sq2": "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE",
    "ic": "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_BUSY_CYCLES",
    "fetch": "FETCH_SIZE",
    "write": "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum",
    # vector-memory path (TA -> TCP(L1) -> TCC(L2)); <= 4 counters of one block per pass
    "tcp1": "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_TCP_LATENCY_sum TCP_TOTAL_ACCESSES_sum",
    "tcp2": "TCP_PENDING_STALL_CYCLES_sum TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCP_TA_TCP_STATE_READ_sum",
    "tcp3": "TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TOTAL_CACHE_ACCESSES_sum",
    "tlb": "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum TCP_UTCL1_STALL_INFLIGHT_MAX_sum",
    "tlb2": "TCP_UTCL1_SERIALIZATION_STALL_sum TCP_UTCL1_STALL_MULTI_MISS_sum TCP_UTCL1_STALL_UTCL2_REQ_OUT_OF_CREDITS_sum TCP_UTCL1_THRASHING_STALL_sum",
    "ta": "TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_TOTAL_WAVEFRONTS_sum",
    "td": "TD_TD_BUSY_sum TD_TC_STALL_sum TD_LOAD_WAVEFRONT_sum GRBM_GUI_ACTIVE",
    "tcc": "TCC_REQ_sum TCC_READ_SECTORS_sum TCC_TAG_STALL_sum TCC_BUSY_avr",
    "tcc2": "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum",
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default="gpurun_out/prof")
    ap.add_argument("--groups", default="sq,fetch,write")
    ap.add_argument("--match", default="")
    ap.add_argument("--pass-timeout", type=int, default=240)
    ap.add_argument("--keep-db", action="store_true", help="keep the rocprofv3 databases (tens of MB) next to summary.txt")
    ap.add_argument("cmd", nargs=argparse.REMAINDER)
    a = ap.parse_args()
    cmd = a.cmd[1:] if a.cmd and a.cmd[0] == "--" else a.cmd
    os.makedirs(a.out, exist_ok=True)
    env = dict(os.environ, TMPDIR="/tmp")
    runs = [("trace", ["--kernel-trace", "--stats"])] + [(g, ["--pmc"] + GROUPS[g].split()) for g in a.groups.split(",") if g]
    for name, flags in runs:
        d = os.path.join(a.out, name)
        with open(os.path.join(a.out, name + ".log"), "w") as log:
            try:    # some counter groups crash rocprofv3 on this image and then hang: bound every pass
                subprocess.call(["rocprofv3"] + flags + ["-d", d, "-o", name, "--"] + cmd, stdout=log, stderr=log, env=env,
                                timeout=a.pass_timeout)
            except subprocess.TimeoutExpired:
                print(f"pass {name}: timed out after {a.pass_timeout} s", file=sys.stderr)
    stats, counters = {}, {}
    for db in glob.glob(os.path.join(a.out, "trace", "**", "*.db"), recursive=True):
        cur = sqlite3.connect(db).cursor()
        for name, calls, total, avg, pct in cur.execute("select name,total_calls,total_duration,average,percentage from top_kernels"):
            stats[name] = (calls, total, avg, pct)
    for g in a.groups.split(","):
        for db in glob.glob(os.path.join(a.out, g, "**", "*.db"), recursive=True):
            cur = sqlite3.connect(db).cursor()
            for kn, cn, v, cnt in cur.execute("select kernel_name,counter_name,avg(value),count(*) from counters_collection group by kernel_name,counter_name"):
                counters.setdefault(kn, {})[cn] = v
    lines = []
    for name, (calls, total, avg, pct) in sorted(stats.items(), key=lambda kv: -kv[1][1]):
        if a.match and a.match not in name:
            continue
        short = name.split("(")[0][-60:]
        lines.append(f"{short}: calls={calls} avg={avg:.2f}us total={total / 1e3:.3f}ms ({pct:.1f}%)")
        c = counters.get(name, {})
        if c:
            parts = []
            for key in sorted(c):
                v = c[key]
                if key == "FETCH_SIZE":
                    parts.append(f"FETCH_SIZE={v * 1024 / 1e6:.1f}MB raw, x2={v * 2048 / 1e6:.1f}MB")      # the counter is in KiB; MB = 10^6 bytes
                elif key == "WRITE_SIZE":
                    parts.append(f"WRITE_SIZE={v * 1024 / 1e6:.1f}MB")
                else:
                    parts.append(f"{key}={v:.4g}")
            lines.append("    " + "  ".join(parts))
    text = "\n".join(lines)
    if not a.keep_db:
        import shutil
        for name, _ in runs:
            shutil.rmtree(os.path.join(a.out, name), ignore_errors=True)
    print(text)
    open(os.path.join(a.out, "summary.txt"), "w").write("cmd: " + " ".join(cmd) + "\n" + text + "\n")


if __name__ == "__main__":
    main()
