"""CPU: build-time audit of the kernels that OWN accumulation registers: the 4 x 64 attention kernel
(csrc/attn.hip::flash_fwd64_kernel) and the four-wave GEMM (csrc/gemm_4w.hip::gemm4w_kernel, a0 .. a255).

That kernel OWNS accumulation registers a64..a255 (O^T and Q^T live there for a whole item, named literally inside asm
statements).  hipcc does not know: a spill, or any v_accvgpr_* it generates itself into that range, would corrupt them without
a fault or a message (MI355X guide 5.7 item 4).  So the device assembly is checked after every build: no scratch, no AGPR
touched outside an asm statement, the register file split leaves room for all 256 accumulation registers, and hipcc has put no
wait for memory (vmcnt) inside the tile loop other than the kernel's own counted ones - the LDS-DMA pieces are asm statements
precisely so that no s_waitcnt vmcnt(0) lands in front of the next LDS read."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def attn_asm(tmp_path_factory):
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("no hipcc")
    from g2vlm_amd import build
    out = tmp_path_factory.mktemp("asm") / "attn.s"
    cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-mllvm", "-amdgpu-kernarg-preload-count=16",
           *build.FILE_FLAGS.get("attn.hip", []), "-I" + os.path.join(ROOT, "include"), "-I" + build.CSRC, "-S", "--cuda-device-only",
           os.path.join(build.CSRC, "attn.hip"), "-o", str(out)]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    return out.read_text()


@pytest.mark.timeout(600)
def test_attn64_owns_its_accumulation_registers(attn_asm):
    s = attn_asm
    name = next(m for m in re.findall(r"^(_Z\S*flash_fwd64_kernel\S*):", s, re.M))
    i = s.index("\n", s.index(name + ":"))
    body = s[i:s.index("s_endpgm", i)]
    assert body.count("v_mfma_f32_32x32x16_bf16") >= 6 * 72
    assert "scratch_" not in body, "the kernel spills"
    in_asm, outside = False, []
    for ln in body.split("\n"):
        if "#ASMSTART" in ln:
            in_asm = True
        elif "#ASMEND" in ln:
            in_asm = False
        elif not in_asm and not ln.strip().startswith(";") and ("accvgpr" in ln or re.search(r"[ ,\[]a\[?\d", ln)):
            outside.append(ln.strip())
    assert not outside, outside[:5]
    desc = re.search(r"\.amdhsa_kernel " + re.escape(name) + r"\n(.*?)\.end_amdhsa_kernel", s, re.S).group(1)
    val = lambda k: int(re.search(r"\." + k + r"\s+(\d+)", desc).group(1))     # noqa: E731
    assert val("amdhsa_private_segment_fixed_size") == 0
    accum, nxt = val("amdhsa_accum_offset"), val("amdhsa_next_free_vgpr")
    assert nxt - accum == 256 and nxt <= 512, (accum, nxt)                        # a0..a255 allocated behind the arch VGPRs
    # inside the steady-state tile (the blocks that issue 8 DMA pieces) the only vmcnt waits are the kernel's own counted vmcnt(8)
    blocks = re.split(r"\n(?=\.LBB\d+_\d+:)", body)
    steady = [b for b in blocks if b.count("global_load_lds_dwordx4") == 8 and b.count("v_mfma_f32_32x32x16_bf16") == 72]
    assert len(steady) == 2, len(steady)                                             # the two ping-pong copies
    for b in steady:
        waits = re.findall(r"s_waitcnt[^\n]*vmcnt\((\d+)\)", b)
        assert waits == ["8"], waits
        n_instr = sum(1 for ln in b.split("\n") if ln.startswith("\t") and not ln.strip().startswith((";", ".")))
        assert n_instr <= 520, n_instr                                               # <= 7.2 instructions per MFMA (8 x 32 form: 10.4)


@pytest.fixture(scope="module")
def gemm4w_asm(tmp_path_factory):
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("no hipcc")
    from g2vlm_amd import build
    out = tmp_path_factory.mktemp("asm4w") / "gemm_4w.s"
    # G2V_4W_FEW: the SwiGLU and fp32-residual epilogues at tile heights 288 / 160 / 128 (six of the 36 instantiations; the
    # main loop is the same template for all of them) - the whole file takes three minutes to compile
    cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-mllvm", "-amdgpu-kernarg-preload-count=16", "-DG2V_4W_FEW",
           *build.FILE_FLAGS.get("gemm_4w.hip", []), "-I" + os.path.join(ROOT, "include"), "-I" + build.CSRC, "-S", "--cuda-device-only",
           os.path.join(build.CSRC, "gemm_4w.hip"), "-o", str(out)]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    return out.read_text()


@pytest.mark.timeout(900)
def test_gemm4w_owns_its_accumulation_registers(gemm4w_asm):
    """gemm4w_kernel names a0 .. a255 literally inside its MFMA asm statements; hipcc must keep every C++ value in VGPRs: no
    scratch, no v_accvgpr_* or a-register operand outside an asm statement, 256 AGPRs allocated, and in the steady two-K-tile
    loop exactly the kernel's own waits (two counted vmcnt, never vmcnt(0)), four barriers, and nothing but the MFMAs, the DMA
    pieces (s_mov m0 / s_nop / load), the fragment reads and <= 40 scalar instructions."""
    s = gemm4w_asm
    names = re.findall(r"^(_Z\S*gemm4w_kernel\S*):", s, re.M)
    assert len(names) == 6, names
    for name in names:
        i = s.index("\n", s.index(name + ":"))
        body = s[i:s.index(".Lfunc_end", i)]
        assert "scratch_" not in body, (name, "the kernel spills")
        in_asm, outside = False, []
        for ln in body.split("\n"):
            if "#ASMSTART" in ln:
                in_asm = True
            elif "#ASMEND" in ln:
                in_asm = False
            elif not in_asm and not ln.strip().startswith(";") and ("accvgpr" in ln or re.search(r"[ ,\[]a\[?\d", ln)):
                outside.append(ln.strip())
        assert not outside, (name, outside[:5])
        desc = re.search(r"\.amdhsa_kernel " + re.escape(name) + r"\n(.*?)\.end_amdhsa_kernel", s, re.S).group(1)
        val = lambda k: int(re.search(r"\." + k + r"\s+(\d+)", desc).group(1))     # noqa: E731
        assert val("amdhsa_private_segment_fixed_size") == 0
        accum, nxt = val("amdhsa_accum_offset"), val("amdhsa_next_free_vgpr")
        assert nxt - accum == 256 and nxt <= 512, (name, accum, nxt)
        mt = sum(int(x) for x in re.search(r"gemm4w_kernelILi\d+ELi(\d)ELi(\d)E", name).groups())
        blocks = re.split(r"\n(?=\.LBB\d+_\d+:)", body)
        steady = [b for b in blocks if b.count("v_mfma_f32_16x16x32_bf16") == 2 * 16 * mt]
        assert len(steady) == 1, (name, len(steady))
        b = steady[0]
        assert b.count("global_load_lds_dwordx4") == 2 * (mt + 8) and b.count("ds_read_b128") == 2 * (2 * mt + 16) and b.count("s_barrier") == 4
        waits = re.findall(r"s_waitcnt[^\n]*vmcnt\((\d+)\)", b)
        ma0 = int(re.search(r"gemm4w_kernelILi\d+ELi(\d)", name).group(1))
        assert waits == [str(ma0 + 4)] * 2, (name, waits)
        n_instr = sum(1 for ln in b.split("\n") if ln.startswith("\t") and not ln.strip().startswith((";", ".", "#")))
        assert n_instr <= 32 * mt + 3 * 2 * (mt + 8) + 2 * (2 * mt + 16) + 40, (name, n_instr)   # MFMAs + 3 per DMA piece + reads + 40


@pytest.mark.timeout(1800)
def test_every_shipped_instantiation_of_the_agpr_owning_kernels_is_spill_free():
    """The two asm audits above look at the attention kernel and at six of the 36 gemm4w instantiations; this one covers the
    shipped build whole: build.py compiles attn.hip and gemm_4w.hip with -Rpass-analysis=kernel-resource-usage and keeps the
    remarks beside the objects.  Every flash_fwd64 / gemm4w kernel: no scratch, all 256 AGPRs allocated (they are the
    accumulators), at most 256 VGPRs, one wave per SIMD."""
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("no hipcc")
    from g2vlm_amd import build
    build.build()
    seen = 0
    for src, pat in (("gemm_4w.hip", "gemm4w_kernel"), ("attn.hip", "flash_fwd64_kernel")):
        txt = open(build.resources_path(src)).read()
        cur = None
        rows = []
        for ln in txt.splitlines():
            m = re.search(r"remark:\s+(?:\S+:\d+:\d+:\s+)?(.*?)\s+\[-Rpass", ln)
            if not m:
                continue
            t = m.group(1)
            if t.startswith("Function Name:"):
                cur = {"name": t.split(":", 1)[1].strip()}
                rows.append(cur)
            elif cur is not None and ":" in t:
                k, v = t.split(":", 1)
                cur[k.strip()] = v.strip()
        mine = [r for r in rows if pat in r["name"]]
        assert len(mine) == (36 if src == "gemm_4w.hip" else 1), (src, len(mine))
        for r in mine:
            assert int(r["ScratchSize [bytes/lane]"]) == 0, r
            assert int(r["AGPRs"]) == 256 and int(r["VGPRs"]) <= 256, r
            assert int(r["Occupancy [waves/SIMD]"]) == 1, r
            seen += 1
    assert seen == 37
