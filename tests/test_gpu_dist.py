"""Data parallelism on the REAL engine: two fresh child processes share the one GPU of the box, each
builds VdsrModel on libsrx.so, attaches dist.py's exchange (gloo: RCCL needs one GPU per rank) and takes two
train steps on its half of a global batch.  Required: both ranks end with bit-identical parameters, and those
equal one process stepping the concatenated batch -- for Adam and for the Momentum + clip path (clip AFTER the
reduce, vdsr/vdsr/model_vdsr.py:170-181 applied to the averaged gradient).  Also: `python bench.py --gpus 2`
started bare launches its own ranks and prints one valid line."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WORKER = os.path.join(ROOT, 'tests', 'dist_gpu_worker.py')

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _run_ranks(out_dir, world, args, timeout=600):
    port = _free_port()
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR='127.0.0.1',
                   MASTER_PORT=str(port), SRX_DIST_BACKEND='gloo', HSA_ENABLE_IPC_MODE_LEGACY='0')
        procs.append(subprocess.Popen([sys.executable, WORKER, str(out_dir)] + [str(a) for a in args], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    outs = []
    for p in procs:
        try:
            out, _ = p.communicate(timeout=timeout)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        outs.append(out.decode(errors='replace'))
    for r, p in enumerate(procs):
        assert p.returncode == 0, 'rank %d failed:\n%s' % (r, outs[r][-3000:])
    return [np.load(os.path.join(str(out_dir), 'rank%d.npz' % r)) for r in range(world)]


def _single(out_dir, args):
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK')}
    subprocess.check_call([sys.executable, WORKER, str(out_dir)] + [str(a) for a in args], env=env)
    return np.load(os.path.join(str(out_dir), 'rank0.npz'))


@pytest.mark.parametrize('use_adam', [1, 0], ids=['adam', 'momentum_clip'])
def test_two_ranks_equal_single_process(tmp_path, use_adam):
    steps, layers, n_global = 2, 6, 8
    # momentum path: the gradients of this problem reach ~7e-3 (bias of the last layer); lr 4 puts the clip bound
    # 0.01/lr = 2.5e-3 inside their range, so that clipping before the reduce (wrong) and after it (right) differ
    lr = 5e-5 if use_adam else 4.0
    args = [use_adam, steps, layers, n_global, lr]
    d2 = tmp_path / 'w2'
    d2.mkdir()
    r0, r1 = _run_ranks(d2, 2, args)
    d1 = tmp_path / 'w1'
    d1.mkdir()
    one = _single(d1, args)                         # same seed as rank 0, whole batch, no hook

    # replicas: identical bit for bit, same step count (rank 1 started from other weights and step 5)
    assert int(r0['global_step']) == int(r1['global_step']) == steps
    for key in ('params', 'opt_m', 'first_grad') + (('opt_v',) if use_adam else ()):
        np.testing.assert_array_equal(r0[key], r1[key], err_msg=key)
    assert int(r0['n_hook_calls']) == steps and float(r0['allreduce_ms']) > 0.0

    # the averaged gradient of the two half batches == the gradient of the whole batch
    g, gref = r0['first_grad'], one['first_grad']
    gmax = np.abs(gref).max()
    assert np.abs(g - gref).max() <= 2e-6 * gmax, np.abs(g - gref).max() / gmax
    if not use_adam:
        cap = 0.01 / lr
        assert np.abs(gref).max() > cap, 'test must exercise the clip'
    # parameters after two optimizer steps
    p, pref = r0['params'], one['params']
    pmax = np.abs(pref).max()
    if use_adam:
        # Adam divides by sqrt(v)+eps: where |g| is of the order of eps=1e-8 the update amplifies the rounding
        # difference of the two summation orders; everywhere else the bound is 1e-6 of the largest parameter
        big = np.abs(gref) > 1e-5 * gmax
        assert np.abs(p - pref)[big].max() <= 1e-6 * pmax
        assert np.abs(p - pref).max() <= 2 * steps * lr       # no element can move further than lr per step
    else:
        # linear in the (clipped) gradient: the rounding difference of the gradients times lr per step
        assert np.abs(p - pref).max() <= 1e-6 * pmax + steps * lr * 2e-6 * gmax
    # the loss of step 1 is the local mean: the mean over ranks is the global one
    np.testing.assert_allclose(0.5 * (r0['losses'][0] + r1['losses'][0]), one['losses'][0], rtol=2e-6)


def test_enet_two_ranks_equal_single_process(tmp_path):
    """EnhanceNet-PAT data parallel (dist.attach_flat): both trainers' gradients averaged over two ranks equal the
    single-process gradients of the concatenated batch (every loss of build_enet is a mean over the batch), replicas
    stay bit-identical, the parameters after one d run + one g run agree."""
    d2 = tmp_path / 'w2'
    d2.mkdir()
    r0, r1 = _run_ranks(d2, 2, ['enet', 4])
    d1 = tmp_path / 'w1'
    d1.mkdir()
    one = _single(d1, ['enet', 4])
    for key in ('g_params', 'd_params', 'g_grad', 'd_grad'):
        np.testing.assert_array_equal(r0[key], r1[key], err_msg=key)
    assert int(r0['global_step']) == int(one['global_step']) == 1
    for key in ('g_grad', 'd_grad'):
        ref = one[key]
        assert np.abs(r0[key] - ref).max() <= 5e-5 * np.abs(ref).max(), (key, np.abs(r0[key] - ref).max() / np.abs(ref).max())
    np.testing.assert_allclose(0.5 * (r0['a_loss'] + r1['a_loss']), one['a_loss'], rtol=1e-5)
    np.testing.assert_allclose(0.5 * (r0['g_loss_all'] + r1['g_loss_all']), one['g_loss_all'], rtol=1e-4)
    # Adam(1e-4): no element moves further than lr per step; where the gradient is not tiny the updates agree
    for pk, gk in (('g_params', 'g_grad'), ('d_params', 'd_grad')):
        assert np.abs(r0[pk] - one[pk]).max() <= 2.1e-4
        big = np.abs(one[gk]) > 1e-3 * np.abs(one[gk]).max()
        assert np.abs(r0[pk] - one[pk])[big].max() <= 2e-6 * max(np.abs(one[pk]).max(), 1.0)


def test_enet_two_ranks_level_a_resumed_rank_zero(tmp_path):
    """dist.attach_flat: rank 0 comes out of a checkpoint (global_step 6 -- a d_trainer step of the schedule,
    enet/enet/experiment_train.py:112 --, both optimizers' slots and step counts), rank 1 found none and sits at step 5
    without slots.  After attach_flat both hold rank 0's state; one d run + one g run later the replicas are
    bit-identical in parameters, slots and counts, and agree with ONE process resumed from the same state."""
    d2 = tmp_path / 'w2'
    d2.mkdir()
    r0, r1 = _run_ranks(d2, 2, ['enet_resumed', 4])
    d1 = tmp_path / 'w1'
    d1.mkdir()
    one = _single(d1, ['enet_resumed', 4])
    for key in ('g_params', 'd_params', 'g_grad', 'd_grad', 'g_m', 'g_v', 'd_m', 'd_v'):
        np.testing.assert_array_equal(r0[key], r1[key], err_msg=key)
    for r in (r0, r1, one):
        assert (int(r['global_step']), int(r['g_t']), int(r['d_t'])) == (7, 7, 3)
    assert int(r0['n_hook_calls']) == 2 and float(r0['allreduce_ms']) > 0
    for key in ('g_grad', 'd_grad'):
        assert np.abs(r0[key] - one[key]).max() <= 5e-5 * np.abs(one[key]).max(), key
    for key in ('g_m', 'd_m', 'g_v', 'd_v'):
        assert np.abs(r0[key] - one[key]).max() <= 1e-5 * np.abs(one[key]).max(), key
    # (started from one's slots: a rank that had begun with EMPTY slots would be off by the whole first moment)
    assert np.abs(one['g_m']).max() > 1e-4


def _group_of_one(out_dir, args):
    """ONE fresh child process that forms a world-size-1 process group with the backend dist.py picks on a GPU box:
    `nccl` = RCCL.  Runs the branches the gloo rehearsals cannot: all_reduce(AVG) on a device buffer, the device-side
    broadcasts of attach() / attach_flat(), destroy_process_group."""
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'SRX_DIST_BACKEND')}
    env.update(RANK='0', LOCAL_RANK='0', WORLD_SIZE='1', MASTER_ADDR='127.0.0.1', MASTER_PORT=str(_free_port()),
               SRX_TEST_GROUP_OF_ONE='1', SRX_TEST_EXPECT_BACKEND='nccl', HSA_ENABLE_IPC_MODE_LEGACY='0')
    p = subprocess.run([sys.executable, WORKER, str(out_dir)] + [str(a) for a in args], env=env, stdout=subprocess.PIPE,
                       stderr=subprocess.STDOUT, timeout=600)
    assert p.returncode == 0, p.stdout.decode(errors='replace')[-3000:]
    return np.load(os.path.join(str(out_dir), 'rank0.npz'))


def test_rccl_group_of_one_vdsr(tmp_path):
    """RCCL initialised by THIS code on the one GPU of the box: init_process_group('nccl'), dist.attach(timed=True) on
    the real VdsrModel, two train steps through all_reduce(AVG) of the flat gradient buffer -- bit-equal to the
    hook-less run (the average over one rank is the identity), hook timed, group destroyed cleanly."""
    args = [1, 2, 6, 8, 5e-5]
    dn = tmp_path / 'nccl'
    dn.mkdir()
    r = _group_of_one(dn, args)
    d1 = tmp_path / 'plain'
    d1.mkdir()
    one = _single(d1, args)
    for key in ('params', 'opt_m', 'opt_v', 'first_grad', 'losses'):
        np.testing.assert_array_equal(r[key], one[key], err_msg=key)
    assert int(r['n_hook_calls']) == 2 and float(r['allreduce_ms']) > 0.0 and int(r['global_step']) == 2


def test_rccl_group_of_one_enet(tmp_path):
    """The same for dist.attach_flat (EnhanceNet's two flat buffers), from a resumed state: device-side broadcast of the
    counts and of both pairs of Adam slots, one all_reduce(AVG) per trainer run."""
    dn = tmp_path / 'nccl'
    dn.mkdir()
    r = _group_of_one(dn, ['enet_resumed', 2])
    d1 = tmp_path / 'plain'
    d1.mkdir()
    one = _single(d1, ['enet_resumed', 2])
    for key in ('g_params', 'd_params', 'g_grad', 'd_grad', 'g_m', 'g_v', 'd_m', 'd_v', 'a_loss', 'g_loss_all'):
        np.testing.assert_array_equal(r[key], one[key], err_msg=key)
    assert int(r['n_hook_calls']) == 2 and float(r['allreduce_ms']) > 0.0
    assert (int(r['global_step']), int(r['g_t']), int(r['d_t'])) == (7, 7, 3)


def test_bench_launches_its_own_ranks():
    """`python bench.py --gpus 2` with no torchrun environment: the parent spawns two fresh ranks (gloo on a
    one-GPU box), rank 0 prints ONE JSON line with the contract's keys."""
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_PORT')}
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '2', '--warmup',
                          '1', '--no-extras'], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
    assert out.returncode == 0, out.stderr.decode(errors='replace')[-3000:]
    lines = [l for l in out.stdout.decode().splitlines() if l.startswith('{')]
    assert len(lines) == 1
    line = json.loads(lines[0])
    assert line['n_gpus'] == 2 and line['steps'] == 2 and line['scaling'] == 'weak'
    assert line['config']['global_batch'] == 512
    assert line['value'] > 0 and line['allreduce_ms'] > 0
    assert line['roofline']['frac'] > 0


def test_bench_default_line_has_the_contract_keys():
    """`python bench.py` (N = 1, as the driver runs it): ONE JSON line with the contract's keys, `roofline` and
    `cpu_baseline` objects, and the secondary north-star numbers."""
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_PORT')}
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--steps', '3', '--warmup', '1'], env=env,
                         stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
    assert out.returncode == 0, out.stderr.decode(errors='replace')[-3000:]
    lines = [l for l in out.stdout.decode().splitlines() if l.startswith('{')]
    assert len(lines) == 1
    line = json.loads(lines[0])
    for key in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling',
                'vs_baseline', 'dtype', 'data', 'config', 'roofline', 'cpu_baseline'):
        assert key in line, key
    assert line['n_gpus'] == 1 and line['steps'] == 3 and line['dtype'] == 'f32' and line['vs_baseline'] is None
    assert 'workload' in line['config'] and 'model' not in line['config']
    r = line['roofline']
    assert r['bound'] == 'mfma' and r['unit'] == 'TFLOP/s' and abs(r['frac'] - r['achieved'] / r['peak']) < 1e-3
    assert 0.5 < r['frac'] < 1.0 and ('traffic' in r)
    # HBM bytes of the dominant kernel: measured by bench.py itself (two rocprofv3 --pmc child passes before it touches the
    # GPU), not replayed from profiles/traffic.json; within a few per cent of the algorithmic bytes (read x once, write y once)
    assert r['traffic_source'].startswith('measured in this run'), r
    assert 0.98 < r['traffic'] / r['traffic_algorithmic'] < 1.15, r
    c = line['cpu_baseline']
    assert c['kind'] == 'port' and c['cores'] >= 1 and c['value'] > 0 and 'sample' in c
    assert abs(line['value'] - 256 * 1000.0 / line['ms_per_step']) < 0.01 * line['value']
    for key in ('dgrad', 'wgrad', 'subpixel', 'espcn_c2_us', 'srcnn_c1_us', 'enet_pat', 'fwd_hr_mpix_per_s',
                'vdsr_recipe_64x128', 'srcnn_c1_cpu_ms', 'espcn_c2_cpu_ms', 'espcn_train_us', 'srcnn_train_ms'):
        assert key in line, key
    rec = line['vdsr_recipe_64x128']
    assert 'error' not in rec, rec
    assert rec['train_ms'] > 0 and 0.3 < rec['train_frac_of_fp32_mfma_peak'] < 1.0
    for k in ('fwd', 'dgrad', 'wgrad'):
        assert 0.3 < rec[k]['frac'] < 1.0, (k, rec[k])
    for k in ('srcnn_c1_cpu_ms', 'espcn_c2_cpu_ms'):
        assert line[k]['port_ms'] > 0 and line[k]['library_ms'] > 0 and line[k]['cores'] >= 1, line[k]
    assert 'error' not in line['espcn_train_us'] and line['espcn_train_us']['graph_replay_us'] > 0, line['espcn_train_us']
    assert 'error' not in line['srcnn_train_ms'] and line['srcnn_train_ms']['train_ms'] > 0, line['srcnn_train_ms']
    assert line['subpixel']['bound'] == 'hbm' and 0.3 < line['subpixel']['frac'] < 1.0
    assert 'error' not in line['enet_pat'], line['enet_pat']
    assert line['enet_pat']['tiles_512']['g_trainer_ms'] > 0
