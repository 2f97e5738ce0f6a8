"""CPU: the host-side bookkeeping of bench.py that the driver's line depends on (no GPU, no engine): the roofline object's round-6 fields
(`top_symbols`, the power-limited matrix peak) from a synthetic per-launch record, and the sizing of the cpu_baseline's whole-host leg."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def _rec():
    # (ms, algorithmic flops, cfg id, (M, N, K), algorithmic bytes): one Winograd launch, two fused tails, one 256-tile, one streaming 1x1
    return [(3.2, 1.66e12, 73, (1404928, 256, 2304), 2.88e9), (0.9, 2.0e11, 70, (1404928, 320, 896), 3.6e9), (0.8, 1.9e11, 70, (1404928, 320, 896), 3.6e9),
            (0.28, 7.2e10, 50, (87808, 256, 1024), 4.5e8), (0.2, 4.6e10, 71, (87808, 1024, 256), 8.1e8)]


def test_roofline_top_symbols_and_power_limited_peak():
    r = bench.roofline_of(_rec(), 'f16x3')
    assert r['bound'] == 'mfma' and r['kernel'].startswith('wino_x3w_kernel') and r['launches_per_step'] == 1
    assert abs(r['frac'] - 1.66e12 / 3.2e-3 / 1e12 / bench.PEAK_BF16_TFLOPS) < 1e-3
    top = r['top_symbols']
    assert [t['kernel'].split('(')[0].split('<')[0].strip() for t in top] == ['wino_x3w_kernel', 'bneck_x3_kernel', 'igemm_dma_kernel']   # by total ms of the step
    assert top[1]['launches'] == 2 and abs(top[1]['ms'] - 1.7) < 1e-9
    for t in top:
        assert set(t) >= {'ms', 'tflops', 'frac', 'matrix_pipe_frac', 'algorithmic_GBps', 'hbm_frac', 'mfma_busy', 'waves_waiting', 'traffic_over_algorithmic', 'share_of_sampled_ms'}
        assert abs(t['frac'] - t['tflops'] / bench.PEAK_BF16_TFLOPS) < 1e-3
    assert abs(top[0]['matrix_pipe_frac'] / top[0]['frac'] - 2.0) < 1e-2      # Winograd F(2,3): two matrix-pipe FLOPs per algorithmic FLOP
    assert abs(top[1]['matrix_pipe_frac'] / top[1]['frac'] - 3.0) < 1e-2      # direct split contraction: three
    assert r['power_limited_matrix_peak_tflops'] == 1700.0 and 'r05_d' in r['power_limited_matrix_peak_source']
    assert abs(r['matrix_pipe_frac_of_power_limited_peak'] - 2.0 * r['achieved'] / 1700.0) < 1e-3
    assert set(r['top_symbols_counter_build_ids']) == {'pmc_mfma', 'pmc_traffic', 'library'}
    json.dumps(r)                                                                 # the object goes into the driver's JSON line


def test_roofline_names_the_fp16_instantiations():
    rec = [(1.0, 8.0e11, 30, (351232, 256, 2304), 1.0e9), (0.5, 2.0e11, 25, (351232, 128, 512), 5.0e8)]
    r = bench.roofline_of(rec, 'f16')
    assert r['kernel'].startswith('igemm_dma_kernel<f16,') and all('<bf16,' not in t['kernel'] for t in r['top_symbols'])
    assert bench.roofline_of(rec, 'bf16')['kernel'].startswith('igemm_dma_kernel<bf16,')


def test_cpu_quota_reads_the_cgroup(tmp_path, monkeypatch):
    real_open = open

    def fake(files):
        def _open(path, *a, **k):
            if path in files:
                if files[path] is None:
                    raise OSError(path)
                p = tmp_path / path.strip('/').replace('/', '_')
                p.write_text(files[path])
                return real_open(p, *a, **k)
            return real_open(path, *a, **k)
        return _open
    monkeypatch.setattr('builtins.open', fake({'/sys/fs/cgroup/cpu.max': '1600000 100000\n'}))
    assert bench.cpu_quota_cores() == 16.0                                        # what the pool's 256-CPU boxes give a container
    monkeypatch.setattr('builtins.open', fake({'/sys/fs/cgroup/cpu.max': 'max 100000\n'}))
    assert bench.cpu_quota_cores() is None
    monkeypatch.setattr('builtins.open', fake({'/sys/fs/cgroup/cpu.max': None, '/sys/fs/cgroup/cpu/cpu.cfs_quota_us': '-1\n', '/sys/fs/cgroup/cpu/cpu.cfs_period_us': '100000\n'}))
    assert bench.cpu_quota_cores() is None
    monkeypatch.setattr('builtins.open', fake({'/sys/fs/cgroup/cpu.max': None, '/sys/fs/cgroup/cpu/cpu.cfs_quota_us': '800000\n', '/sys/fs/cgroup/cpu/cpu.cfs_period_us': '100000\n'}))
    assert bench.cpu_quota_cores() == 8.0
