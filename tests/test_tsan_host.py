"""The C ABI's threading promise (include/srx.h: concurrent calls from several host threads), checked with a
ThreadSanitizer build of the HOST code only -- a CPU test.  This file and its driver are listed in .gpurunignore:
sanitizer builds must not travel to the GPU pool, and the GPU run does not need them."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(not os.path.exists('/opt/rocm/bin/hipcc'), reason='needs hipcc')
def test_abi_host_code_is_thread_safe_under_tsan(tmp_path):
    """include/srx.h promises concurrent calls from several host threads.  The host side of srx_api.hip is built
    with ThreadSanitizer (CPU build only: the kernel launchers are stubbed out) and two threads hammer
    srx_conv2d_workspace_bytes / srx_set_conv_path / a failing srx_conv2d_fwd (thread-local error text)."""
    exe = str(tmp_path / 'tsan_abi')
    csrc = os.path.join(ROOT, 'ml_super_resolution_amd', 'csrc')
    cmd = ['/opt/rocm/bin/hipcc', '-O1', '-g', '-std=c++17', '-fsanitize=thread', '--cuda-host-only', '--offload-arch=gfx950',
           '-Wno-unused-result', os.path.join(csrc, 'srx_api.hip'), '-x', 'hip', os.path.join(ROOT, 'tests', 'tsan_abi_driver.cpp'),
           '-o', exe, '-lpthread']
    build = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    assert build.returncode == 0, build.stdout.decode(errors='replace')[-4000:]
    run = subprocess.run([exe], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=300,
                         env=dict(os.environ, TSAN_OPTIONS='halt_on_error=1 exitcode=66'))
    out = run.stdout.decode(errors='replace')
    assert run.returncode == 0 and 'ThreadSanitizer' not in out and 'tsan driver ok' in out, out[-4000:]
