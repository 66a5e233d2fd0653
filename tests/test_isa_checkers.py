"""The two ISA-level checkers `make check` / build() run over the kernels' assembly (scripts/check_asm_hazards.py,
scripts/check_async_regs.py) must themselves catch what they are there for: fed hand-written assembly fragments."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run(script, text, tmp_path, name):
    p = tmp_path / name
    p.write_text(text)
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'scripts', script), str(p)], capture_output=True, text=True)
    return r.returncode, r.stdout


MFMA_BLOCK = """\t;;#ASMSTART
\tv_mfma_f32_16x16x4_f32 v[0:3], a0, v10, v[0:3]
\tv_mfma_f32_16x16x4_f32 v[4:7], a0, v11, v[4:7]
\t;;#ASMEND
"""


def test_hazard_checker_flags_valu_write_feeding_an_mfma_block(tmp_path):
    clean = "k:\n\tds_read_b128 v[10:13], v20\n\ts_waitcnt lgkmcnt(0)\n" + MFMA_BLOCK
    rc, out = run('check_asm_hazards.py', clean, tmp_path, 'clean.s')
    assert rc == 0 and '1 asm MFMA blocks checked, 0 hazards' in out
    # a VALU write of an operand register right in front of the block: 2 wait states are missing
    bad = "k:\n\tv_mov_b32_e32 v11, v30\n" + MFMA_BLOCK
    rc, out = run('check_asm_hazards.py', bad, tmp_path, 'bad.s')
    assert rc == 1 and '1 hazards' in out
    # ... also when it writes an accumulator, and one instruction earlier
    bad2 = "k:\n\tv_mov_b32_e32 v5, v30\n\ts_add_i32 s0, s0, 1\n" + MFMA_BLOCK
    rc, out = run('check_asm_hazards.py', bad2, tmp_path, 'bad2.s')
    assert rc == 1
    # a block that starts with its own s_nop (the GUARD variants) is exempt
    guarded = "k:\n\tv_mov_b32_e32 v11, v30\n" + MFMA_BLOCK.replace(";;#ASMSTART\n", ";;#ASMSTART\n\ts_nop 1\n")
    rc, out = run('check_asm_hazards.py', guarded, tmp_path, 'guarded.s')
    assert rc == 0


LOAD = """\t;;#ASMSTART
\ts_mov_b64 exec, s[10:11]
\tbuffer_load_dwordx4 v[40:43], v9, s[4:7], s12 offen
\ts_mov_b64 exec, -1
\t;;#ASMEND
"""
CONSUME = """\t;;#ASMSTART
\ts_waitcnt vmcnt(0)
\tds_write_b128 v8, v[40:43]
\t;;#ASMEND
"""


def test_async_register_checker(tmp_path):
    ok = "k:\n" + LOAD + "\tv_add_u32_e32 v8, s0, v7\n\tds_read_b128 v[50:53], v20\n" + CONSUME
    rc, out = run('check_async_regs.py', ok, tmp_path, 'ok.s')
    assert rc == 0 and '1 asm loads checked, 0 violations' in out
    # the compiler copies the destination before the consumer: it would copy data that has not arrived
    copy = "k:\n" + LOAD + "\tv_mov_b32_e32 v60, v41\n" + CONSUME
    rc, out = run('check_async_regs.py', copy, tmp_path, 'copy.s')
    assert rc == 1 and 'reads the destination' in out
    # the same through the accumulation-register file (value parked in AGPRs, moved to a VGPR too early)
    agpr = ("k:\n" + LOAD.replace('v[40:43]', 'a[8:11]') + "\tv_accvgpr_read_b32 v60, a9\n" +
            CONSUME.replace('v[40:43]', 'a[8:11]'))
    rc, out = run('check_async_regs.py', agpr, tmp_path, 'agpr.s')
    assert rc == 1
    # a wait statement that merely NAMES the registers (in an asm comment) counts as the consumer
    named = "k:\n" + LOAD + "\t;;#ASMSTART\n\ts_waitcnt vmcnt(0) ; v40 v41 v42 v43\n\t;;#ASMEND\n\tv_mov_b32_e32 v60, v41\n"
    rc, out = run('check_async_regs.py', named, tmp_path, 'named.s')
    assert rc == 0
    # a compiler-emitted instruction that only REDEFINES the register (an exclusive, textually later path) ends the scan
    redef = "k:\n" + LOAD + "\tbuffer_load_dwordx4 v[40:43], v9, s[4:7], 0 offen\n\tv_mov_b32_e32 v60, v41\n"
    rc, out = run('check_async_regs.py', redef, tmp_path, 'redef.s')
    assert rc == 0
