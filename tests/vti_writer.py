"""Test utility: writes depth-map .vti files in every data mode of the VTK XML ImageData format, from the format's
published description (not from VTK code): ascii, inline binary (base64), appended raw / base64, with or without the
zlib compressor, UInt32 / UInt64 headers, little / big endian.  Used to exercise host/vti_reader.cpp."""
import base64
import struct
import zlib

import numpy as np

_TYPES = {np.dtype("float64"): "Float64", np.dtype("float32"): "Float32", np.dtype("uint8"): "UInt8",
          np.dtype("int32"): "Int32"}


def _payload(raw: bytes, header: str, endian: str, compress: bool, block: int, b64: bool) -> bytes:
    fmt = endian + ("Q" if header == "UInt64" else "I")
    if not compress:
        unit = struct.pack(fmt, len(raw)) + raw
        return base64.b64encode(unit) if b64 else unit
    blocks = [raw[i:i + block] for i in range(0, len(raw), block)]
    comp = [zlib.compress(b) for b in blocks]
    last = len(blocks[-1]) if blocks and len(blocks[-1]) != block else 0
    head = struct.pack(fmt, len(blocks)) + struct.pack(fmt, block) + struct.pack(fmt, last)
    head += b"".join(struct.pack(fmt, len(c)) for c in comp)
    body = b"".join(comp)
    # base64: the header is its own unit (padded), the blocks follow as a second unit
    return base64.b64encode(head) + base64.b64encode(body) if b64 else head + body


def write_vti(path, arrays, width, height, mode="ascii", compress=False, header="UInt32", big_endian=False,
              block=32768, origin=(0, 0, 0), spacing=(1, 1, 1)):
    """arrays: dict name -> ndarray of shape [H, W] or [H, W, C] (vtk row order: row 0 = bottom image row).
    mode: ascii | binary | appended-raw | appended-base64."""
    endian = ">" if big_endian else "<"
    attrs = f'type="ImageData" version="{"1.0" if header == "UInt64" else "0.1"}" byte_order="{"BigEndian" if big_endian else "LittleEndian"}"'
    if header == "UInt64":
        attrs += ' header_type="UInt64"'
    if compress:
        attrs += ' compressor="vtkZLibDataCompressor"'
    ext = f"0 {width - 1} 0 {height - 1} 0 0"
    out = [b'<?xml version="1.0"?>\n', f"<VTKFile {attrs}>\n".encode(),
           f'  <ImageData WholeExtent="{ext}" Origin="{origin[0]} {origin[1]} {origin[2]}" '
           f'Spacing="{spacing[0]} {spacing[1]} {spacing[2]}">\n'.encode(),
           f'    <Piece Extent="{ext}">\n      <PointData Scalars="Depths">\n'.encode()]
    appended = []
    offset = 0
    for name, a in arrays.items():
        a = np.ascontiguousarray(a)
        comps = a.shape[2] if a.ndim == 3 else 1
        tname = _TYPES[a.dtype]
        tag = f'        <DataArray type="{tname}" Name="{name}" NumberOfComponents="{comps}" '
        raw = a.astype(a.dtype.newbyteorder(endian)).tobytes()
        if mode == "ascii":
            vals = " ".join(repr(float(v)) if a.dtype.kind == "f" else str(int(v)) for v in a.reshape(-1))
            out.append((tag + 'format="ascii">\n          ' + vals + "\n        </DataArray>\n").encode())
        elif mode == "binary":
            out.append((tag + 'format="binary">\n          ').encode() + _payload(raw, header, endian, compress, block, True)
                       + b"\n        </DataArray>\n")
        else:
            p = _payload(raw, header, endian, compress, block, mode == "appended-base64")
            out.append((tag + f'format="appended" offset="{offset}"/>\n').encode())
            appended.append(p)
            offset += len(p)
    out.append(b"      </PointData>\n      <CellData>\n      </CellData>\n    </Piece>\n  </ImageData>\n")
    if appended:
        enc = "base64" if mode == "appended-base64" else "raw"
        out.append(f'  <AppendedData encoding="{enc}">\n   _'.encode() + b"".join(appended) + b"\n  </AppendedData>\n")
    out.append(b"</VTKFile>\n")
    with open(path, "wb") as f:
        f.write(b"".join(out))
