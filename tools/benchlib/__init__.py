"""Helpers of bench.py (the driver contract lives in bench.py itself: launcher, ranks, the timed headline loop, the one JSON line).

    benchlib.cpu_baselines   the cpu_baseline leg (C oracle over the host cores, torch sparse-mm restatement, neural oracle)
    benchlib.neural          configs[2] and the configs[3] / configs[4] shards, per-kernel MFMA rooflines, the training step
    benchlib.secondary       what the default run measures next to the headline; the flat summary the driver's record keeps
"""
import os

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
MFMA_F32_PEAK_TFLOPS = 157.3   # MI355X_MICROARCH.md: fp32-input MFMA = the fp32 vector peak
N_SIMD, CLOCK_HZ = 1024, 2.4e9  # 256 CUs x 4 SIMDs, 2.4 GHz


def algorithmic_bytes_per_iteration(E, V, F):
    "SURVEY.md section 8(d): streaming model of one SP iteration, 41 B/edge + 36 B/variable + 8 B/clause"
    return 41 * E + 36 * V + 8 * F


def grouped():
    """Does this process join a torch.distributed group?  Always with several ranks; with ONE rank only on request (PDP_DIST_FORCE=1 under
    torch.distributed.run --nproc-per-node 1): the barrier and the two all-reduces then go through RCCL on a one-GPU box exactly as they
    do on eight, and the line says rccl_ranks = 1 with the backend that ran."""
    return int(os.environ.get('WORLD_SIZE', '1')) > 1 or (os.environ.get('PDP_DIST_FORCE') == '1' and 'RANK' in os.environ)
