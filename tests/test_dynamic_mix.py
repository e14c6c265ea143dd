"""scripts/dynamic_mix.py: the attribution of ISA instructions to CTR_MARK segments (host logic of the roofline, no GPU)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "scripts"))

ASM = """_ZN12_GLOBAL__N_113render_kernelILj107EEEvNS_5KArgsEPfS2_S2_Py: ; @kernel
; %bb.0:
	;;#ASMSTART
	; CTR_MARK 0
	;;#ASMEND
	s_load_dwordx4 s[0:3], s[4:5], 0x0
	v_mov_b32_e32 v1, s0
	v_fma_f32 v2, v1, v1, v1
.LBB0_1:                                ; a loop body whose mark was sunk below its arithmetic
	v_mul_f32_e32 v3, v2, v2
	v_rcp_f32_e32 v4, v3
	;;#ASMSTART
	; CTR_MARK 0x41
	;;#ASMEND
	s_add_u32 s6, s6, 1
	s_cbranch_scc1 .LBB0_1
; %bb.2:                                ; no mark: continues segment 0x41
	v_add_f32_e32 v5, v4, v4
	;;#ASMSTART
	; CTR_MARK 2
	;;#ASMEND
	v_cmp_gt_f32_e32 vcc, v5, v4
	s_endpgm
; Occupancy: 8
"""


def test_block_level_attribution(tmp_path):
    import dynamic_mix
    p = tmp_path / "k.s"
    p.write_text(ASM)
    seg = dynamic_mix.segments(dynamic_mix.kernel_body(str(p), "107"))
    # block 0: mark 0 owns the load, the SGPR move (half rate) and the fma (full rate)
    assert seg[0]["smem"] == 1 and seg[0]["H"] == 1 and seg[0]["F"] == 1 and seg[0]["copies"] == 1
    # block 1 belongs to mark 0x41 entirely, although the comment sits below the multiply and the reciprocal
    assert seg[0x41]["F"] == 1 and seg[0x41]["Q"] == 1 and seg[0x41]["salu"] == 1 and seg[0x41]["branch"] == 1
    # block 2: what precedes its first mark goes to that mark; the compare (half rate) too
    assert seg[2]["F"] == 1 and seg[2]["H"] == 1


def test_pricing_and_cross_check(tmp_path):
    p = tmp_path / "k.s"
    p.write_text(ASM)
    counts = tmp_path / "c.json"
    counts.write_text(json.dumps({"workload": "w", "per_launch": {"0": 10, "65": 1000, "2": 10}}))
    pmc = tmp_path / "p.json"
    pmc.write_text(json.dumps({"valu_insts_per_launch": 2040.0, "salu_insts_per_launch": 1000.0, "smem_insts_per_launch": 10.0}))
    out = tmp_path / "o.json"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "dynamic_mix.py"), str(p), "107", str(counts), "--pmc", str(pmc),
                        "--out", str(out)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    m = json.loads(out.read_text())
    # 10 x (F + H) + 1000 x (F + Q) + 10 x (F + H) = 2040 VALU; cycles 20 x (2.2 + 4.1) + 1000 x (2.2 + 8.1)
    assert m["valu_insts_modelled"] == 2040
    assert abs(m["mean_issue_cycles_per_valu"] - (20 * 6.3 + 1000 * 10.3) / 2040) < 1e-3
    assert m["pmc_check"]["modelled_over_measured_valu"] == 1.0 and m["pmc_check"]["modelled_over_measured_smem"] == 1.0
