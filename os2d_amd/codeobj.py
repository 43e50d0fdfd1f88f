"""Kernel resource figures of a built HIP library, read from its embedded code objects (no toolchain needed).

    python -m os2d_amd.codeobj [path/to/libos2d_hip.so] [substring ...]

The host ELF carries one clang offload bundle per translation unit in its ``.hip_fatbin`` section; every bundle holds an
amdgcn ELF whose NT_AMDGPU_METADATA note (msgpack) lists, per kernel, the register counts, the SPILLED registers, the
scratch bytes per work item and the static LDS size.  tests/test_codeobj.py pins ``.vgpr_spill_count == 0`` for every
kernel of the default head path (the transform kernels shipped with 31 / 20 spilled registers)."""
import os
import struct
import sys

MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"


def _sections(data):
    """{name: (offset, size)} of a little-endian ELF64 image."""
    if data[:4] != b"\x7fELF" or data[4] != 2:
        raise ValueError("not an ELF64 image")
    shoff, = struct.unpack_from("<Q", data, 0x28)
    shentsize, shnum, shstrndx = struct.unpack_from("<HHH", data, 0x3A)
    heads = [struct.unpack_from("<IIQQQQIIQQ", data, shoff + i * shentsize) for i in range(shnum)]
    stroff = heads[shstrndx][4]
    out = {}
    for h in heads:
        end = data.index(b"\0", stroff + h[0])
        out.setdefault(data[stroff + h[0]:end].decode(), []).append((h[4], h[5], h[1]))
    return out


def device_images(path, arch="gfx950"):
    """The amdgcn code objects (bytes) of every bundle in the library's .hip_fatbin section that target ``arch``."""
    with open(path, "rb") as f:
        data = f.read()
    images = []
    for off, size, _ in _sections(data).get(".hip_fatbin", []):
        blob = data[off:off + size]
        pos = blob.find(MAGIC)
        while pos >= 0:
            n, = struct.unpack_from("<Q", blob, pos + len(MAGIC))
            p = pos + len(MAGIC) + 8
            for _ in range(n):
                eoff, esize, tlen = struct.unpack_from("<QQQ", blob, p)
                triple = blob[p + 24:p + 24 + tlen].decode()
                p += 24 + tlen
                if arch in triple and esize:
                    images.append(blob[pos + eoff:pos + eoff + esize])
            pos = blob.find(MAGIC, pos + len(MAGIC))
    return images


def _notes(image):
    for off, size, sh_type in [s for v in _sections(image).values() for s in v]:
        if sh_type != 7:        # SHT_NOTE
            continue
        p = off
        while p + 12 <= off + size:
            namesz, descsz, ntype = struct.unpack_from("<III", image, p)
            p += 12
            name = image[p:p + namesz].rstrip(b"\0")
            p += (namesz + 3) & ~3
            desc = image[p:p + descsz]
            p += (descsz + 3) & ~3
            yield name, ntype, desc


def kernels(path, arch="gfx950"):
    """{kernel symbol: {vgprs, agprs, sgprs, vgpr_spills, sgpr_spills, scratch_bytes, lds_bytes, max_threads}} of every kernel."""
    import msgpack
    out = {}
    for image in device_images(path, arch):
        for name, ntype, desc in _notes(image):
            if name != b"AMDGPU" or ntype != 32:     # NT_AMDGPU_METADATA
                continue
            meta = msgpack.unpackb(desc, raw=False, strict_map_key=False)
            for k in meta.get("amdhsa.kernels", []):
                out[k[".name"]] = {"vgprs": k.get(".vgpr_count"), "agprs": k.get(".agpr_count", 0), "sgprs": k.get(".sgpr_count"),
                                   "vgpr_spills": k.get(".vgpr_spill_count", 0), "sgpr_spills": k.get(".sgpr_spill_count", 0),
                                   "scratch_bytes": k.get(".private_segment_fixed_size", 0),
                                   "lds_bytes": k.get(".group_segment_fixed_size", 0), "max_threads": k.get(".max_flat_workgroup_size")}
    return out


def main(argv):
    here = os.path.dirname(os.path.abspath(__file__))
    path = argv[1] if len(argv) > 1 and os.path.exists(argv[1]) else os.path.join(here, "lib", "libos2d_hip.so")
    pats = [a for a in argv[1:] if a != path and a != "--names"]
    ks = kernels(path)
    if "--names" in argv:        # the sorted kernel symbols, one per line: tests/golden/kernels.txt is this output
        print("\n".join(sorted(ks)))
        return 0
    print("{:>5} {:>5} {:>6} {:>7} {:>7}  kernel".format("vgpr", "agpr", "spills", "scratch", "lds"))
    for name in sorted(ks):
        if pats and not any(p in name for p in pats):
            continue
        k = ks[name]
        print("{:>5} {:>5} {:>6} {:>7} {:>7}  {}".format(k["vgprs"], k["agprs"], k["vgpr_spills"], k["scratch_bytes"], k["lds_bytes"], name))
    return 0


if __name__ == "__main__":
    sys.exit(main(sys.argv))
