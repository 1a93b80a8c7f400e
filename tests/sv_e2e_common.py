"""Shared by the CPU and GPU tests of BASELINE configs[3] (`tiddit --sv --skip_assembly`): regenerates the synthetic WGS-shaped
BAM + FASTA a golden fixture (tests/golden/sv_e2e*.json, made by tests/golden/make_golden.py from the compiled reference) was
computed on.  Everything is derived from the seeds stored in the fixture, so the file is identical on every machine."""
import json
import os


def load_fixture(golden_dir, name):
    return json.load(open(os.path.join(golden_dir, name)))


def materialise(fixture, directory, threads=8):
    """-> (bam path, fasta path, contigs)"""
    from tiddit_amd import synth_bam
    P = fixture["params"]
    contigs = synth_bam.contigs_for(P)
    fa, bam = os.path.join(directory, "ref.fa"), os.path.join(directory, "WGS.bam")
    seqs = synth_bam.write_fasta(fa, contigs, seed=P["fasta_seed"])
    info = synth_bam.write_wgs_sv_bam(bam, contigs, depth=P["depth"], read_len=P["read_len"], insert=P["insert"], insert_sd=P["insert_sd"],
                                      seed=P["seed"], sv_per_mb=P["sv_per_mb"], threads=threads, ref_seqs=seqs)
    assert info["n_records"] == fixture["n_records"] and info["events"] == fixture["events"]
    return bam, fa, contigs
