"""`make check-isa` (stereoreconstruction_amd/csrc/check_isa.py, part of the library's default build target): the checker of
what geodesic_dma_kernel asks of the compiler catches what it is there to catch -- on hand-made instruction streams."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHECK = os.path.join(ROOT, "stereoreconstruction_amd", "csrc", "check_isa.py")

REQ4 = ["s_mov_b32 m0, s19", "s_nop 0", "global_load_lds_dwordx4 v[6:7], off"]
REQ1 = ["s_mov_b32 m0, s4", "s_nop 0", "global_load_lds_dword v[0:1], off"]
PAD = ["v_add_f64 v[0:1], v[2:3], v[4:5]"] * 600


def _run(tmp_path, body):
    f = tmp_path / "k.s"
    f.write_text("_ZN3srh19geodesic_dma_kernelILi5EEEvPKNS_7ViewDevEiPKd10srh_paramsiiPdS7_:\n" +
                 "\n".join("\t" + t for t in body + ["s_endpgm"]) + "\n")
    r = subprocess.run([sys.executable, CHECK, str(f)], capture_output=True, text=True)
    return r.returncode, r.stdout


def _good():
    return (REQ4 + REQ1 + ["s_waitcnt vmcnt(0)", "ds_read_b64 v[0:1], v2"] + PAD + ["s_barrier"] + REQ4 + REQ1 + PAD +
            ["s_waitcnt vmcnt(0)", "s_cbranch_scc1 .LBB0_1", "s_waitcnt vmcnt(0)"])


def test_the_shape_the_kernel_has_passes(tmp_path):
    rc, out = _run(tmp_path, _good())
    assert rc == 0, out


def test_a_compiler_value_in_m0_is_caught(tmp_path):
    body = _good()
    body.insert(700, "s_mov_b32 m0, -1")                  # the compiler parks something of its own in m0
    rc, out = _run(tmp_path, body)
    assert rc == 1 and "A: m0 is touched outside" in out, out
    body = _good()
    body.insert(700, "v_readlane_b32 s5, v7, m0")
    rc, out = _run(tmp_path, body)
    assert rc == 1 and "A:" in out, out


def test_a_hand_over_before_the_wait_is_caught(tmp_path):
    body = REQ4 + REQ1 + ["s_waitcnt vmcnt(0)", "ds_read_b64 v[0:1], v2"] + PAD + ["s_barrier"] + REQ4 + REQ1 + PAD + ["s_barrier", "s_waitcnt vmcnt(0)"]
    rc, out = _run(tmp_path, body)
    assert rc == 1 and "B:" in out, out


def test_a_read_before_the_first_tile_has_landed_is_caught(tmp_path):
    body = REQ4 + REQ1 + ["ds_read_b64 v[0:1], v2", "s_waitcnt vmcnt(0)"] + PAD + ["s_barrier"] + REQ4 + REQ1 + PAD + ["s_waitcnt vmcnt(0)"]
    rc, out = _run(tmp_path, body)
    assert rc == 1 and "C:" in out, out
