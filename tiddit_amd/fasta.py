"""Minimal indexed FASTA access (the reference uses pysam.FastaFile, tiddit_gc.pyx:7-15).

Reads a whole contig as a uint8 numpy array (newlines stripped with one reshape) from a ``.fai``
index; builds the index when it is missing (what ``pysam.faidx`` does at __main__.py:95-97)."""
import os

import numpy as np


def build_fai(path):
    """Python reference of tdt_fasta_write_fai (csrc/tdt_format.hip), kept for the CPU tests"""
    entries = []
    with open(path, "rb") as f:
        name = None
        length = offset = linebases = linewidth = 0
        pos = 0
        for line in f:
            ln = len(line)
            if line.startswith(b">"):
                if name is not None:
                    entries.append((name, length, offset, linebases, linewidth))
                name = line[1:].split()[0].decode()
                length = 0
                offset = pos + ln
                linebases = linewidth = 0
            elif name is not None:
                bases = len(line.rstrip(b"\r\n"))
                if linebases == 0 and bases:
                    linebases, linewidth = bases, ln
                length += bases
            pos += ln
        if name is not None:
            entries.append((name, length, offset, linebases, linewidth))
    with open(path + ".fai", "w") as out:
        for e in entries:
            out.write("%s\t%d\t%d\t%d\t%d\n" % e)
    return entries


class FastaFile:
    def __init__(self, path):
        self.path = path
        if not os.path.isfile(path + ".fai"):
            from . import _native
            tmp = "%s.fai.%d.tmp" % (path, os.getpid())          # N ranks may all find the index missing: each writes its own file and
            _native.check(_native.load().tdt_fasta_write_fai(path.encode(), tmp.encode()))   # build_fai() in C
            os.replace(tmp, path + ".fai")                       # the rename is atomic — nobody ever parses a half-written index
        self.index = {}
        self.references = []
        for line in open(path + ".fai"):
            c = line.rstrip("\n").split("\t")
            self.index[c[0]] = tuple(int(v) for v in c[1:5])
            self.references.append(c[0])

    def get_reference_length(self, contig):
        return self.index[contig][0]

    def fetch_array(self, contig):
        """whole contig -> uint8[length] (bases exactly as in the file, case preserved)"""
        length, offset, linebases, linewidth = self.index[contig]
        if length == 0:
            return np.zeros(0, dtype=np.uint8)
        nfull = length // linebases
        tail = length - nfull * linebases
        nbytes = nfull * linewidth + tail
        with open(self.path, "rb") as f:
            f.seek(offset)
            raw = np.frombuffer(f.read(nbytes), dtype=np.uint8)
        if linewidth == linebases:
            return raw[:length].copy()
        if raw.size < nfull * linewidth:
            # the last contig may end on a full line without a final line end (tail == 0 at EOF)
            raw = np.concatenate([raw, np.zeros(nfull * linewidth - raw.size, dtype=np.uint8)])
        body = raw[:nfull * linewidth].reshape(nfull, linewidth)[:, :linebases].reshape(-1)
        return np.concatenate([body, raw[nfull * linewidth:nfull * linewidth + tail]])

    def fetch_raw(self, contig):
        """the contig's bytes exactly as they are in the file (line ends included) -> (uint8[nbytes], length, linebases, linewidth);
        the GC kernel reads this layout directly (csrc/tdt_gc.hip: gc_fasta_bins), so nothing is stripped or copied on the host"""
        length, offset, linebases, linewidth = self.index[contig]
        if length == 0:
            return np.zeros(0, dtype=np.uint8), 0, linebases, linewidth
        nfull = length // linebases if linebases else 0
        tail = length - nfull * linebases
        nbytes = nfull * linewidth + tail
        if tail == 0:
            nbytes -= linewidth - linebases
        return np.fromfile(self.path, dtype=np.uint8, count=nbytes, offset=offset), length, linebases, linewidth

    def fetch(self, contig, start=0, end=None):
        a = self.fetch_array(contig)
        return a[start:end].tobytes().decode()
