"""Which kernels does libaggf.so contain?  Read from the library's own dynamic symbol table (no external tool): every
`__global__` function leaves a host stub `__device_stub__<name>` and a kernel handle `<name>`; the handle is what a
launch passes to the runtime and what the library's launch counter (aggf_coverage_dump) reports."""
import re
import struct


def dynamic_symbols(path):
    """Names of the defined symbols of an ELF64 little-endian shared object's .dynsym."""
    with open(path, "rb") as fh:
        data = fh.read()
    assert data[:4] == b"\x7fELF" and data[4] == 2 and data[5] == 1, "not an ELF64 LE file"
    shoff, = struct.unpack_from("<Q", data, 0x28)
    shentsize, shnum = struct.unpack_from("<HH", data, 0x3A)
    sections = []
    for i in range(shnum):
        off = shoff + i * shentsize
        _name, stype, _flags, _addr, soff, ssize, link, _info, _align, entsize = struct.unpack_from("<IIQQQQIIQQ", data, off)
        sections.append((stype, soff, ssize, link, entsize))
    names = []
    for stype, soff, ssize, link, entsize in sections:
        if stype != 11:  # SHT_DYNSYM
            continue
        _t, stroff, strsize, _l, _e = sections[link]
        for k in range(ssize // entsize):
            st_name, _info, _other, shndx, _value, _size = struct.unpack_from("<IBBHQQ", data, soff + k * entsize)
            if shndx == 0 or st_name == 0:
                continue
            end = data.index(b"\0", stroff + st_name)
            names.append(data[stroff + st_name:end].decode())
    return names


_STUB = re.compile(r"(\d+)__device_stub__")


def compiled_kernels(path):
    """Mangled handle names of every kernel in the library."""
    out = set()
    for n in dynamic_symbols(path):
        m = _STUB.search(n)
        if m:
            out.add(n[:m.start()] + str(int(m.group(1)) - len("__device_stub__")) + n[m.end():])
    return out


def demangle(names):
    """Best effort (c++filt / llvm-cxxfilt if present), else the mangled names."""
    import shutil
    import subprocess

    tool = shutil.which("c++filt") or shutil.which("llvm-cxxfilt") or "/opt/rocm/lib/llvm/bin/llvm-cxxfilt"
    names = list(names)
    try:
        out = subprocess.run([tool], input="\n".join(names), capture_output=True, text=True, check=True).stdout.splitlines()
        if len(out) == len(names):
            return dict(zip(names, out))
    except Exception:
        pass
    return {n: n for n in names}
