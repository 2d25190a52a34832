#!/usr/bin/env python
"""PMC probe of the hot kernels (GPU box). Runs tools/microbench.py under rocprofv3 once per counter group (counters that this
rocprofv3 does not list are dropped), summarises the rocpd database per kernel and deletes it.
usage: pmc_probe.py <out_prefix> [pretrain] [steps] [variant]"""
import os
import re
import shutil
import subprocess
import sys

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GROUPS = {
    "sq": ["SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VMEM", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR"],
    "sq2": ["SQ_WAVES", "SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS", "SQ_INST_CYCLES_VMEM", "SQ_WAIT_INST_LDS"],
    "tcp": ["TCP_TOTAL_CACHE_ACCESSES_sum", "TCP_TCC_READ_REQ_sum", "TCP_TCC_ATOMIC_WITHOUT_RET_REQ_sum", "TCP_PENDING_STALL_CYCLES_sum"],
    "tcp2": ["TCP_TOTAL_ACCESSES_sum", "TCP_TCC_WRITE_REQ_sum", "TCP_TCP_TA_DATA_STALL_CYCLES_sum", "TCP_TA_TCP_STATE_READ_sum"],
    "ta": ["TA_BUSY_avr", "TA_TA_BUSY_sum", "TA_ADDR_STALLED_BY_TC_CYCLES_sum", "TA_DATA_STALLED_BY_TC_CYCLES_sum"],
    "ta2": ["TA_FLAT_READ_WAVEFRONTS_sum", "TA_FLAT_ATOMIC_WAVEFRONTS_sum", "TA_FLAT_WAVEFRONTS_sum", "TA_BUFFER_WAVEFRONTS_sum"],
    "tcc": ["TCC_HIT_sum", "TCC_MISS_sum", "TCC_REQ_sum", "TCC_ATOMIC_sum"],
    "tcc2": ["TCC_EA0_RDREQ_sum", "TCC_EA0_WRREQ_sum", "TCC_EA0_ATOMIC_sum", "TCC_READ_sum"],
    "grbm": ["GRBM_GUI_ACTIVE", "GRBM_COUNT"],
    # memory-side request sizes (calibrates FETCH_SIZE / WRITE_SIZE, which tally every request at 64 B: MI355X_MICROARCH.md "HBM")
    "rdsize": ["TCC_EA0_RDREQ_sum", "TCC_EA0_RDREQ_32B_sum", "TCC_EA0_RDREQ_64B_sum", "TCC_EA0_RDREQ_128B_sum"],
    "wrsize": ["TCC_EA0_WRREQ_sum", "TCC_EA0_WRREQ_64B_sum", "TCC_EA0_ATOMIC_sum"],
    # MFMA utilisation (north_star: "evidenced by rocprof HBM GB/s and MFMA utilisation"): busy cycles of the matrix pipe vs the SQ's, MFMA instruction / op counts
    "mfma": ["SQ_VALU_MFMA_BUSY_CYCLES", "SQ_BUSY_CYCLES", "SQ_INSTS_VALU_MFMA_MOPS_F16", "SQ_INSTS_MFMA", "SQ_WAVE_CYCLES", "SQ_INSTS_VALU"],
}
KERNELS = "k_train_fused|k_train_fwd_bwd|k_grad_bin|k_grad_accumulate|k_inference|k_optimizer|k_wgrad|k_compute_loss_v2|k1_count|k1_write|k1_setup"


def main():
    out = sys.argv[1]
    pretrain = sys.argv[2] if len(sys.argv) > 2 else "300"
    steps = sys.argv[3] if len(sys.argv) > 3 else "6"
    variant = sys.argv[4] if len(sys.argv) > 4 else "default"
    only = sys.argv[5].split(",") if len(sys.argv) > 5 else None
    os.chdir("/tmp")
    env = dict(os.environ, TMPDIR="/tmp")
    avail = subprocess.run(["rocprofv3", "-L"], capture_output=True, text=True, env=env).stdout
    names = set(re.findall(r"\b[A-Z][A-Za-z0-9_]{3,}\b", avail))
    with open(out + "_available.txt", "w") as f:
        f.write("\n".join(sorted(n for n in names if re.match(r"(SQ|TCP|TCC|TA|TD|GRBM|FETCH|WRITE|SPI|CPC|Mfma|VALU|Occup|MemUnit|L2)", n))))
    with open(out + "_summary.txt", "w") as summ:
        for g, ctrs in GROUPS.items():
            if only and g not in only:
                continue
            ok = [c for c in ctrs if c in names]
            summ.write(f"== group {g}: {' '.join(ok)}   (dropped: {' '.join(c for c in ctrs if c not in names)})\n")
            if not ok:
                continue
            d = f"/tmp/pmc_{g}"
            shutil.rmtree(d, ignore_errors=True)
            cmd = ["timeout", "300", "rocprofv3", "--kernel-trace", "--pmc"] + ok + ["--kernel-include-regex", KERNELS, "-d", d, "-o", "p", "--",
                   "python", os.path.join(R, "tools", "microbench.py"), pretrain, steps, variant]
            r = subprocess.run(cmd, capture_output=True, text=True, env=env)
            db = os.path.join(d, "p_results.db")
            if not os.path.exists(db):
                summ.write("   FAILED: " + r.stderr[-600:] + "\n")
                continue
            s = subprocess.run([sys.executable, os.path.join(R, "tools", "rocpd_pmc.py"), db], capture_output=True, text=True)
            summ.write(s.stdout + s.stderr[-300:] + "\n")
            summ.flush()
            shutil.rmtree(d, ignore_errors=True)
    print(open(out + "_summary.txt").read()[-6000:])


if __name__ == "__main__":
    main()
