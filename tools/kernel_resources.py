#!/usr/bin/env python3
"""Register / scratch / LDS budget of every kernel in the built libldpc_hip.so, read from the code-object metadata
(`llvm-readelf --notes` of each gfx950 code object embedded in the library's clang offload bundles).

    python tools/kernel_resources.py [--lib PATH] [--match SUBSTR] [--json]

Used by tests/test_host_cpu.py::test_simulate_kernels_do_not_spill (no GPU needed)."""
import argparse
import json
import os
import re
import struct
import subprocess
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DEFAULT_LIB = os.path.join(ROOT, "ldpc_decoders_amd", "csrc", "libldpc_hip.so")
READELF = "/opt/rocm/lib/llvm/bin/llvm-readelf"
CXXFILT = "c++filt"
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"


def code_objects(blob):
    """Yield the device code objects (bytes) of every uncompressed clang offload bundle found in `blob`."""
    pos = 0
    while True:
        at = blob.find(MAGIC, pos)
        if at < 0:
            return
        pos = at + len(MAGIC)
        (count,) = struct.unpack_from("<Q", blob, pos)
        p = pos + 8
        if count > 64:
            continue
        for _ in range(count):
            off, size, tlen = struct.unpack_from("<QQQ", blob, p)
            triple = blob[p + 24:p + 24 + tlen].decode(errors="replace")
            p += 24 + tlen
            if "amdgcn" in triple and size:
                yield triple, blob[at + off:at + off + size]


def kernels_of(lib=DEFAULT_LIB):
    """-> {demangled kernel name: dict(vgpr, sgpr, spill, scratch, lds, agpr)}"""
    blob = open(lib, "rb").read()
    out = {}
    for _, co in code_objects(blob):
        with tempfile.NamedTemporaryFile(suffix=".co") as fp:
            fp.write(co)
            fp.flush()
            txt = subprocess.run([READELF, "--notes", fp.name], capture_output=True, text=True).stdout
        for block in re.split(r"\n\s*- \.agpr_count:", txt)[1:]:
            block = ".agpr_count:" + block

            def field(name, cast=int, b=block):
                m = re.search(r"\.%s:\s*(\S+)" % re.escape(name), b)
                return cast(m.group(1)) if m else None

            name = field("name", str)
            if not name:
                continue
            out[name.strip("'\"")] = dict(vgpr=field("vgpr_count"), sgpr=field("sgpr_count"), agpr=field("agpr_count"), spill=field("vgpr_spill_count"),
                                          sgpr_spill=field("sgpr_spill_count"), scratch=field("private_segment_fixed_size"),
                                          lds=field("group_segment_fixed_size"))
    names = list(out)
    if names:
        dem = subprocess.run([CXXFILT], input="\n".join(names), capture_output=True, text=True).stdout.splitlines()
        out = {d.replace("ldpc::(anonymous namespace)::", "").replace("void ", ""): out[m] for d, m in zip(dem, names)}
    return out


def short_name(name):
    """Demangled kernel symbol -> the name rocprofv3 / ldpc_decoder_kernel_name report: namespaces, `void ` and the argument list cut."""
    name = name.replace("ldpc::(anonymous namespace)::", "").replace("ldpc::", "").replace("void ", "")
    depth = 0
    for i, ch in enumerate(name):
        depth += ch == "<"
        depth -= ch == ">"
        if ch == "(" and depth == 0:
            return name[:i]
    return name


def _elf_functions(co):
    """(mangled name, machine-code bytes) of every function symbol of one ELF64 code object, and {name: kernel-descriptor bytes}."""
    shoff, = struct.unpack_from("<Q", co, 0x28)
    shentsize, shnum, _ = struct.unpack_from("<HHH", co, 0x3A)
    secs = [struct.unpack_from("<IIQQQQIIQQ", co, shoff + i * shentsize) for i in range(shnum)]  # name type flags addr off size link info align entsize
    funcs, kds = [], {}
    for sec in secs:
        if sec[1] != 2:  # SHT_SYMTAB
            continue
        stroff = secs[sec[6]][4]
        for j in range(sec[5] // 24):
            st_name, st_info, _, st_shndx, st_value, st_size = struct.unpack_from("<IBBHQQ", co, sec[4] + j * 24)
            if st_shndx == 0 or st_shndx >= shnum or st_size == 0:
                continue
            end = co.index(b"\0", stroff + st_name)
            nm = co[stroff + st_name:end].decode(errors="replace")
            host = secs[st_shndx]
            body = co[host[4] + st_value - host[3]:host[4] + st_value - host[3] + st_size]
            if (st_info & 15) == 2:       # STT_FUNC
                funcs.append((nm, body))
            elif nm.endswith(".kd"):      # kernel descriptor (register counts, LDS size, ...)
                kds[nm[:-3]] = body
    return funcs, kds


def kernel_code_hashes(lib=DEFAULT_LIB):
    """{kernel name as rocprofv3 prints it: first 16 hex digits of sha256(machine code + kernel descriptor)} for every kernel of the built
    library -- what profiles/roofline_counters.json records beside a kernel's PMC counters, so that bench.py can tell when the kernel it
    times is no longer the kernel the counters were collected on.  The descriptor's kernel_code_entry_byte_offset (bytes 16..23: where the
    code sits relative to the descriptor) is left out: it moves whenever ANOTHER kernel of the same translation unit is added or changes
    size, with this kernel's instructions and resources untouched.  Pure file parsing: no GPU, no HIP call."""
    import hashlib

    blob = open(lib, "rb").read()
    raw = {}
    for _, co in code_objects(blob):
        if co[:4] != b"\x7fELF":
            continue
        funcs, kds = _elf_functions(co)
        for nm, body in funcs:
            if nm in kds:
                kd = kds[nm]
                raw[nm] = hashlib.sha256(body + kd[:16] + b"\0" * 8 + kd[24:]).hexdigest()[:16]
    names = list(raw)
    if not names:
        return {}
    try:
        dem = subprocess.run([CXXFILT], input="\n".join(names), capture_output=True, text=True).stdout.splitlines()
    except OSError:
        dem = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-cxxfilt"], input="\n".join(names), capture_output=True, text=True).stdout.splitlines()
    return {short_name(d): raw[m] for d, m in zip(dem, names)}


def lib_sha256(lib=DEFAULT_LIB):
    import hashlib

    return hashlib.sha256(open(lib, "rb").read()).hexdigest()


OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"


def m0_report(lib=DEFAULT_LIB):
    """The LDS stores of the fused kernels address through M0 (ds_write_addtid_b32) and set it with inline asm, which the compiler cannot
    see.  That is sound as long as the compiler itself never uses M0 in those kernels.  Disassembles every code object that contains
    such stores and returns (number of addtid stores, M0 references that are NOT one of our `s_mov_b32 m0, sN`, instructions that would
    consume M0 behind our back: movrel / sendmsg / LDS-DMA / GWS)."""
    blob = open(lib, "rb").read()
    stores, foreign, consumers = 0, [], []
    for _, co in code_objects(blob):
        with tempfile.NamedTemporaryFile(suffix=".co") as fp:
            fp.write(co)
            fp.flush()
            txt = subprocess.run([OBJDUMP, "-d", fp.name], capture_output=True, text=True).stdout
        if "ds_write_addtid_b32" not in txt:
            continue
        for line in txt.splitlines():
            ins = line.split("//")[0].strip()
            if "ds_write_addtid_b32" in ins:
                stores += 1
            elif re.search(r"\bm0\b", ins) and not re.match(r"s_mov_b32 m0, s\d+$", ins):
                foreign.append(ins)
            if re.search(r"movrel|s_sendmsg|_lds_|ds_gws|buffer_load.* lds|global_load_lds", ins):
                consumers.append(ins)
    return stores, foreign, consumers


def flat_report(lib=DEFAULT_LIB):
    """flat_load / flat_store instructions in the code objects of the fused kernels (those with ds_write_addtid_b32 stores).  The words
    the waves of a frame hand each other live in the LDS; reached through a generic pointer they become flat accesses (aperture check,
    both counters to wait for) on the store -> barrier -> load chain that ends every sweep -- measured 3-10 % of the kernel time."""
    blob = open(lib, "rb").read()
    found = []
    for _, co in code_objects(blob):
        with tempfile.NamedTemporaryFile(suffix=".co") as fp:
            fp.write(co)
            fp.flush()
            txt = subprocess.run([OBJDUMP, "-d", fp.name], capture_output=True, text=True).stdout
        if "ds_write_addtid_b32" not in txt:
            continue
        found += [ln.split("//")[0].strip() for ln in txt.splitlines() if re.search(r"\bflat_(load|store)_", ln)]
    return found


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--lib", default=DEFAULT_LIB)
    ap.add_argument("--match", default="")
    ap.add_argument("--json", action="store_true")
    ap.add_argument("--hashes", action="store_true", help="print the per-kernel code hashes (kernel_code_hashes) instead of the resources")
    a = ap.parse_args()
    if a.hashes:
        for k, v in sorted(kernel_code_hashes(a.lib).items()):
            if a.match in k:
                print(v, k)
        raise SystemExit(0)
    ks = {k: v for k, v in kernels_of(a.lib).items() if a.match in k}
    if a.json:
        print(json.dumps(ks, indent=1))
    else:
        for k, v in sorted(ks.items()):
            print("%-110s vgpr %3s agpr %3s sgpr %3s spill %3s scratch %4s B lds %6s B" % (k.split("(")[0][:110], v["vgpr"], v["agpr"], v["sgpr"], v["spill"], v["scratch"], v["lds"]))
