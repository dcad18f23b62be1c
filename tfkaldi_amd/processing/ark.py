"""Kaldi binary ark / scp I/O with the interface of the reference's processing/ark.py.

On-disk format (reference processing/ark.py:59-94 reader, :190-211 writer), per matrix:
    <utt_id bytes>                      written with NO separating space (ark.py:204)
    \\0 'B' <'F' | 'D'> 'M' ' '         5 bytes, the scp offset points at the \\0 (ark.py:72, 205-206)
    \\x04 <int32 rows> \\x04 <int32 cols>  little endian (ark.py:80-81, 207-208)
    rows * cols float32 ('F') or float64 ('D'), row-major (ark.py:83-90, 209)
scp line: "<utt_id> <ark path>:<offset>" (ark.py:51-54, 210).  Compressed ('C') and text archives are
rejected the way the reference does: message + exit(1) (ark.py:73-78).
"""
import struct
import sys

import numpy as np

_HEADER = struct.Struct("<xcccc")
_DIM = struct.Struct("<bi")


class ArkReader(object):
    """Reads matrices addressed by an scp file; keeps a cursor (`scp_position`) with wrap-around."""

    def __init__(self, scp_path):
        self.scp_position = 0
        self.utt_ids = []
        self.scp_data = []
        self._files = {}
        with open(scp_path, "r") as fid:
            for line in fid:
                utt_id, path_pos = line.rstrip("\n").split(" ")
                path, pos = path_pos.split(":")
                self.utt_ids.append(utt_id)
                self.scp_data.append((path, pos))

    def _handle(self, path):
        fh = self._files.get(path)
        if fh is None:
            fh = self._files[path] = open(path, "rb")
        return fh

    def read_utt_data(self, index):
        """the matrix of scp entry `index` (float32 or float64, as stored)"""
        path, pos = self.scp_data[index]
        fh = self._handle(path)
        fh.seek(int(pos), 0)
        binary, kind, _, _ = _HEADER.unpack(fh.read(5))
        if binary != b"B":
            print("Input .ark file is not binary")
            sys.exit(1)
        if kind == b"C":
            print("Input .ark file is compressed")
            sys.exit(1)
        _, rows = _DIM.unpack(fh.read(5))
        _, cols = _DIM.unpack(fh.read(5))
        dtype = np.float32 if kind == b"F" else np.float64
        data = np.frombuffer(fh.read(rows * cols * np.dtype(dtype).itemsize), dtype=dtype)
        return data.reshape(rows, cols)

    def read_next_utt(self):
        """(utt_id, matrix, looped): `looped` is True on the read that wrapped to the first entry."""
        if len(self.scp_data) == 0:
            return None, None, True
        looped = self.scp_position >= len(self.scp_data)
        if looped:
            self.scp_position = 0
        self.scp_position += 1
        return self.utt_ids[self.scp_position - 1], self.read_utt_data(self.scp_position - 1), looped

    def read_next_scp(self):
        """advance the cursor and return the utterance id without touching the archive"""
        if self.scp_position >= len(self.scp_data):
            self.scp_position = 0
        self.scp_position += 1
        return self.utt_ids[self.scp_position - 1]

    def read_previous_scp(self):
        """move the cursor back by one; returns the id the cursor pointed at BEFORE the move (the
        reference's behaviour, ark.py:136-150, which BatchDispenser.return_batch relies on)"""
        if self.scp_position < 0:
            self.scp_position = len(self.scp_data) - 1
        self.scp_position -= 1
        return self.utt_ids[self.scp_position + 1]

    def read_utt(self, utt_id):
        return self.read_utt_data(self.utt_ids.index(utt_id))

    def split(self):
        """Drop what has been read so far.  As in the reference (ark.py:161-165) the LAST entry is dropped
        too (`[scp_position:-1]`) and the cursor is left where it was."""
        self.scp_data = self.scp_data[self.scp_position:-1]
        self.utt_ids = self.utt_ids[self.scp_position:-1]

    def close(self):
        for fh in self._files.values():
            fh.close()
        self._files = {}


class ArkWriter(object):
    """Appends float32 matrices to an ark file and lists them in an scp file."""

    def __init__(self, scp_path, default_ark):
        self.scp_path = scp_path
        self.scp_file_write = open(self.scp_path, "w")
        self.default_ark = default_ark

    def write_next_utt(self, utt_id, utt_mat, ark_path=None):
        ark = ark_path or self.default_ark
        mat = np.ascontiguousarray(utt_mat, dtype=np.float32)
        rows, cols = mat.shape
        with open(ark, "ab") as fh:
            fh.write(utt_id.encode() if isinstance(utt_id, str) else utt_id)
            pos = fh.tell()
            fh.write(b"\0BFM ")
            fh.write(_DIM.pack(4, rows))
            fh.write(_DIM.pack(4, cols))
            fh.write(mat.tobytes())
        self.scp_file_write.write("%s %s:%s\n" % (utt_id, ark, pos))

    def close(self):
        self.scp_file_write.close()
