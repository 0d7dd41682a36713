"""CPU: the C-ABI library loads and exports exactly what include/lfi.h declares (no compute without a GPU)."""
import ctypes
import os
import re
import subprocess

import pytest


def _declared_symbols(root):
    text = open(os.path.join(root, "include", "lfi.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(lfi_[a-z0-9_]+)\s*\(", text)))


def test_header_and_binding_agree(native):
    from conftest import ROOT
    declared = _declared_symbols(ROOT)
    assert declared == sorted(native.ABI_SYMBOLS)


def test_library_exports_every_declared_symbol(native):
    from conftest import ROOT
    lib = ctypes.CDLL(native.build.HIP_LIB)
    for name in _declared_symbols(ROOT):
        assert hasattr(lib, name), name
    lib.lfi_abi_version.restype = ctypes.c_int
    assert lib.lfi_abi_version() == 1


def test_header_compiles_as_plain_c(tmp_path):
    from conftest import ROOT
    src = tmp_path / "t.c"
    src.write_text('#include "lfi.h"\nint main(void){ lfi_params p; (void)p; return LFI_ABI_VERSION - 1; }\n')
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Werror", "-pedantic", "-I", os.path.join(ROOT, "include"), "-c", str(src), "-o",
                           str(tmp_path / "t.o")])


def test_code_object_is_gfx950(native):
    # the fat binary inside the shared library must carry gfx950 code objects and no other GPU target
    data = open(native.build.HIP_LIB, "rb").read()
    targets = set(re.findall(rb"amdgcn-amd-amdhsa--(gfx[0-9a-z]+)", data))
    assert targets == {b"gfx950"}, targets


def test_no_gpu_means_error_not_fallback(native):
    lib = native.load_hip_library()
    if lib.lfi_device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(native.LfiError, match="no CPU fallback"):
        native.Context(0)


def test_product_never_touches_the_oracle():
    """oracle/ is test infrastructure: nothing under lfinterpolator_amd/ or include/ may reference it."""
    from conftest import ROOT
    bad = []
    for base in ("lfinterpolator_amd", "include"):
        for dirpath, _, files in os.walk(os.path.join(ROOT, base)):
            for f in files:
                if f.endswith((".py", ".h", ".hpp", ".cpp", ".hip", "Makefile")):
                    text = open(os.path.join(dirpath, f), errors="ignore").read()
                    if re.search(r"lfi_oracle|lfo_[a-z]|from oracle|import oracle|oracle/", text):
                        # comments that merely NAME the oracle are fine; includes / imports / calls are not
                        if re.search(r'#include\s+"[^"]*oracle|import\s+oracle|from\s+oracle|lfo_[a-z0-9_]+\s*\(', text):
                            bad.append(os.path.join(dirpath, f))
    assert not bad, bad


def _gfx950_code_object(native, tmp_path):
    """The gfx950 ELF inside the shared library's clang offload bundle."""
    import struct
    data = open(native.build.HIP_LIB, "rb").read()
    i = data.find(b"__CLANG_OFFLOAD_BUNDLE__")
    assert i >= 0
    n = struct.unpack_from("<Q", data, i + 24)[0]
    off = i + 32
    for _ in range(n):
        o, sz, tl = struct.unpack_from("<QQQ", data, off)
        off += 24
        triple = data[off:off + tl]
        off += tl
        if b"gfx950" in triple:
            path = tmp_path / "gfx950.co"
            path.write_bytes(data[i + o:i + o + sz])
            return str(path)
    raise AssertionError("no gfx950 code object in the bundle")


LLVM_BIN = "/opt/rocm/lib/llvm/bin"


@pytest.mark.skipif(not os.path.exists(os.path.join(LLVM_BIN, "llvm-readelf")), reason="ROCm LLVM tools not installed")
def test_pipelined_kernels_use_no_scratch_and_the_counted_stores(native, tmp_path):
    """The LDS-DMA pipelines (blend_p3, blend_stdx, blend_planar, blend_persist, blend_wave) wait with HAND-COUNTED `s_waitcnt vmcnt(n)`: n is the
    number of vector-memory instructions the kernel itself issued after the DMA it waits for.  Anything the compiler adds to that
    queue behind our back breaks the count silently: register spills (scratch loads / stores are vector-memory operations), or an
    epilogue whose stores were merged or split.  So: none of these kernels may use scratch, and blend_p3's epilogues must consist of
    six 16-byte stores and its DMA issue sites of three LDS loads per octet of images (the compiler may clone a site, never change its size)."""
    co = _gfx950_code_object(native, tmp_path)
    notes = subprocess.run([os.path.join(LLVM_BIN, "llvm-readelf"), "--notes", co], capture_output=True, text=True, check=True).stdout
    kernels = {}
    name = None
    for line in notes.splitlines():
        line = line.strip()
        if line.startswith(".name:"):
            name = line.split(":", 1)[1].strip()
            kernels[name] = {}
        elif name and ":" in line and line.split(":")[0] in (".private_segment_fixed_size", ".vgpr_spill_count", ".sgpr_spill_count", ".vgpr_count"):
            kernels[name][line.split(":")[0]] = int(line.split(":")[1])
    # focus_range_t (round 5): hand-counted lgkmcnt waits on its LDS reads (no scratch), and at most 136 registers per lane — three of its
    # waves per SIMD must leave room for one wave of focus_flagged (at most 104), or the flagged passes queue behind the persistent kernel
    range_t = [k for k in kernels if "focus_range_tI" in k]
    assert len(range_t) == 2, sorted(kernels)
    for k in range_t:
        assert kernels[k][".private_segment_fixed_size"] == 0 and kernels[k][".vgpr_spill_count"] == 0 and kernels[k][".vgpr_count"] <= 136, (k, kernels[k])
    flagged = [k for k in kernels if "focus_flagged" in k]
    assert len(flagged) == 1 and kernels[flagged[0]][".vgpr_count"] <= 104, (flagged, kernels[flagged[0]] if flagged else None)
    pipelined = [k for k in kernels if any(s in k for s in ("blend_p3", "blend_planar", "blend_persist", "blend_wave", "blend_stdx", "blend_afs"))]
    assert sum("blend_afsI" in k for k in pipelined) == 2     # all-focus STD, every sample gathered once: three and four chunks
    assert sum("blend_stdxI" in k for k in pipelined) == 8    # fixed focus: one to four chunks of images, RGBA and planar views
    assert sum("blend_stdxaI" in k for k in pipelined) == 8   # all-focus: one to four chunks, RGBA and planar views
    assert len(pipelined) >= 10, sorted(kernels)
    # <true, chunks, ablation, view groups per wave, view passes, RGBA epilogue>
    p3_name = re.compile(r"blend_p3ILb1ELi(\d)ELi(\d)ELi(\d)ELi(\d)ELb([01])EEEv")
    n_two_groups = 0
    for k in pipelined:
        assert kernels[k][".private_segment_fixed_size"] == 0 and kernels[k][".vgpr_spill_count"] == 0, (k, kernels[k])
        m = p3_name.search(k)
        if m and m.group(3) == "2" and m.group(2) in "02":
            # two waves of 32 views per workgroup: built to run ONE wave per SIMD (192 accumulators) — and it must need more than half the
            # register file, or the hardware could put both workgroups of a CU on the same two SIMDs
            assert 256 < kernels[k][".vgpr_count"] <= 512, (k, kernels[k])
            n_two_groups += 1
        elif not m or m.group(3) == "1":
            assert kernels[k][".vgpr_count"] <= 256, (k, kernels[k])      # two waves per SIMD: two workgroups per CU
    assert n_two_groups >= 3
    dis = subprocess.run([os.path.join(LLVM_BIN, "llvm-objdump"), "-d", co], capture_output=True, text=True, check=True).stdout
    body = {}
    cur = None
    for line in dis.splitlines():
        m = re.match(r"^[0-9a-f]+ <(\S+)>:", line)
        if m:
            cur = m.group(1)
            body[cur] = []
        elif cur:
            body[cur].append(line)
    rgba_out = [k for k in body if p3_name.search(k) and p3_name.search(k).group(2) == "0" and p3_name.search(k).group(5) == "1"]
    assert len(rgba_out) >= 3, sorted(body)  # the RGBA epilogue (round 4): two to four chunks of images at least
    for k in rgba_out:
        text = "\n".join(body[k])
        n_st16 = len(re.findall(r"global_store_dwordx4", text))
        assert n_st16 >= 8 and n_st16 % 8 == 0, (k, n_st16)  # two adjacent 16-byte stores per view (the ragged right edge: dword stores)
        assert "scratch_" not in text and "buffer_store" not in text and "buffer_load" not in text, k
    shipped = [k for k in body if p3_name.search(k) and p3_name.search(k).group(2) == "0" and p3_name.search(k).group(5) == "0"]
    assert len(shipped) >= 5, sorted(body)   # one chunk: one pass / up to four; two to four chunks: two waves of 32 views
    assert sum(p3_name.search(k).group(4) == "4" for k in shipped) == 1     # up to four view passes: one chunk of images, 16 views per wave
    for k in shipped:
        text = "\n".join(body[k])
        if p3_name.search(k).group(1) == "1" and p3_name.search(k).group(3) == "1" and p3_name.search(k).group(4) == "1":
            # the one-pass kernel lets all its fetches land between its last MFMA and its first store (profiles/r03_notes.md §11: −2 % at
            # config 2, −7 … −15 % per rank of config 4; round 2 had this wait by accident, a clean-up lost it once)
            tail = text[text.rindex("v_mfma_f32_16x16x32_f16"):]
            assert "s_waitcnt vmcnt(0)" in tail[:tail.index("global_store_dwordx4")], k
        if p3_name.search(k).group(4) == "4":
            # the pass loop holds no load the compiler would have to wait for behind predicated stores (round 2: s_waitcnt vmcnt(0) per pass)
            loop = text[text.index("s_barrier", text.index("s_barrier") + 1):]
            assert "global_load_dwordx4" not in loop, k
        n_st, n_dma = len(re.findall(r"global_store_dwordx4", text)), len(re.findall(r"global_load_lds_dwordx4", text))
        assert n_st >= 6 and n_st % 6 == 0 and len(re.findall(r"global_store_", text)) == n_st, (k, n_st)
        assert n_dma >= 18 and n_dma % 3 == 0, (k, n_dma)   # an octet of images = three LDS-DMA instructions (R, G, B planes)
        assert "scratch_" not in text and "buffer_store" not in text and "buffer_load" not in text, k
