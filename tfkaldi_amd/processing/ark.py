"""Kaldi binary ark / scp I/O with the interface of the reference's processing/ark.py.

On-disk format (reference processing/ark.py:59-94 reader, :190-211 writer), per matrix:
    <utt_id bytes>                      written with NO separating space (ark.py:204)
    \\0 'B' <'F' | 'D'> 'M' ' '         5 bytes, the scp offset points at the \\0 (ark.py:72, 205-206)
    \\x04 <int32 rows> \\x04 <int32 cols>  little endian (ark.py:80-81, 207-208)
    rows * cols float32 ('F') or float64 ('D'), row-major (ark.py:83-90, 209)
scp line: "<utt_id> <ark path>:<offset>" (ark.py:51-54, 210).  Compressed ('C') and text archives are
rejected the way the reference does: message + exit(1) (ark.py:73-78).
"""
import os
import struct
import sys

import numpy as np

_HEADER = struct.Struct("<xcccc")
_DIM = struct.Struct("<bi")
_F32, _F64 = np.dtype(np.float32), np.dtype(np.float64)


class ArkReader(object):
    """Reads matrices addressed by an scp file; keeps a cursor (`scp_position`) with wrap-around.

    The reference opens, seeks and reads the archive once per utterance and copies the bytes twice (ark.py:59-94).
    Here every archive is opened once, the 15 header bytes of an entry are parsed the first time it is touched and
    cached (`entry`: descriptor, data offset, rows, cols, dtype), and the matrix body is ONE positioned read -- into a
    fresh array (`read_utt_data`) or straight into a caller's batch buffer (`read_into`), which is what the batch
    dispenser's packed path uses."""

    def __init__(self, scp_path):
        self.scp_position = 0
        self.utt_ids = []
        self.scp_data = []
        self._fds = {}
        self._entries = []   # per scp line: None or (fd, data offset, rows, cols, dtype)
        self._lookup = None  # utterance id -> index, built on the first read_utt
        self.bytes_read = 0  # matrix bytes fetched from the archives (tests: a rank touches only its own utterances)
        with open(scp_path, "r") as fid:
            for line in fid:
                utt_id, path_pos = line.rstrip("\n").split(" ")
                path, pos = path_pos.split(":")
                self.utt_ids.append(utt_id)
                self.scp_data.append((path, pos))
        self._entries = [None] * len(self.scp_data)

    def _fd(self, path):
        fd = self._fds.get(path)
        if fd is None:
            fd = self._fds[path] = os.open(path, os.O_RDONLY)
        return fd

    def entry(self, index):
        """(descriptor, offset of the matrix body, rows, cols, dtype) of scp entry `index`; reads the 15-byte header
        on the first call only"""
        ent = self._entries[index]
        if ent is None:
            path, pos = self.scp_data[index]
            fd = self._fd(path)
            head = os.pread(fd, 15, int(pos))
            binary, kind, _, _ = _HEADER.unpack(head[:5])
            if binary != b"B":
                print("Input .ark file is not binary")
                sys.exit(1)
            if kind == b"C":
                print("Input .ark file is compressed")
                sys.exit(1)
            _, rows = _DIM.unpack(head[5:10])
            _, cols = _DIM.unpack(head[10:15])
            ent = self._entries[index] = (fd, int(pos) + 15, rows, cols, _F32 if kind == b"F" else _F64)
        return ent

    def read_into(self, index, out):
        """the matrix body of entry `index` into `out`, a C-contiguous array of the entry's dtype and size"""
        fd, offset, rows, cols, dtype = self.entry(index)
        nbytes = rows * cols * dtype.itemsize
        if out.dtype != dtype or out.nbytes != nbytes or not out.flags.c_contiguous:
            raise ValueError("read_into: the buffer does not match the %d x %d %s matrix" % (rows, cols, dtype))
        # one positioned read in the common case; a read may legally return fewer bytes than asked for (network / FUSE file
        # systems, signals, the kernel's 0x7ffff000 cap per call): carry on where it stopped, and only an END OF FILE is an error
        view, got = memoryview(out).cast("B"), 0
        while got < nbytes:
            n = os.preadv(fd, [view[got:]], offset + got)
            if n <= 0:
                raise IOError("unexpected end of %s: %d of %d bytes of the matrix" % (self.scp_data[index][0], got, nbytes))
            got += n
        self.bytes_read += nbytes

    def read_utt_data(self, index):
        """the matrix of scp entry `index` (float32 or float64, as stored)"""
        _, _, rows, cols, dtype = self.entry(index)
        out = np.empty((rows, cols), dtype=dtype)
        self.read_into(index, out)
        return out

    def _advance(self):
        looped = self.scp_position >= len(self.scp_data)
        if looped:
            self.scp_position = 0
        self.scp_position += 1
        return self.scp_position - 1, looped

    def read_next_utt(self):
        """(utt_id, matrix, looped): `looped` is True on the read that wrapped to the first entry."""
        if len(self.scp_data) == 0:
            return None, None, True
        index, looped = self._advance()
        return self.utt_ids[index], self.read_utt_data(index), looped

    def next_entry(self):
        """advance the cursor like read_next_utt but fetch nothing: (index, utt_id, looped)"""
        index, looped = self._advance()
        return index, self.utt_ids[index], looped

    def read_next_scp(self):
        """advance the cursor and return the utterance id without touching the archive"""
        return self.utt_ids[self._advance()[0]]

    def read_previous_scp(self):
        """move the cursor back by one; returns the id the cursor pointed at BEFORE the move (the
        reference's behaviour, ark.py:136-150, which BatchDispenser.return_batch relies on)"""
        if self.scp_position < 0:
            self.scp_position = len(self.scp_data) - 1
        self.scp_position -= 1
        return self.utt_ids[self.scp_position + 1]

    def read_utt(self, utt_id):
        if self._lookup is None:  # (first occurrence wins, as list.index does in the reference: ark.py:159)
            self._lookup = {}
            for index, key in enumerate(self.utt_ids):
                self._lookup.setdefault(key, index)
        if utt_id not in self._lookup:
            raise ValueError("%r is not in list" % (utt_id,))
        return self.read_utt_data(self._lookup[utt_id])

    def split(self):
        """Drop what has been read so far.  As in the reference (ark.py:161-165) the LAST entry is dropped
        too (`[scp_position:-1]`) and the cursor is left where it was."""
        self.scp_data = self.scp_data[self.scp_position:-1]
        self.utt_ids = self.utt_ids[self.scp_position:-1]
        self._entries = self._entries[self.scp_position:-1]
        self._lookup = None

    def close(self):
        for fd in self._fds.values():
            os.close(fd)
        self._fds = {}
        self._entries = [None] * len(self.scp_data)

    def __del__(self):
        try:
            self.close()
        except Exception:  # interpreter shutdown
            pass


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
