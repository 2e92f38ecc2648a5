"""Runs oracle.nlp.kkt_certificate_ipopt for many (problem, solution) pairs on the host cores (spawned workers: the
parent of a GPU test has the HIP runtime loaded and must not be forked)."""
import multiprocessing as mp
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _one(args):
    if ROOT not in sys.path:
        sys.path.insert(0, ROOT)
    from oracle import nlp
    prob, X, U, s = args
    return nlp.kkt_certificate_ipopt(prob, X, U, s)


def certify(items, procs=None):
    """items: list of (nlp.Problem, X, U, s) -> list of certificate dicts (same order)."""
    items = list(items)
    if not items:
        return []
    avail = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    procs = max(1, min(procs or 16, avail, len(items)))
    if procs == 1:
        return [_one(a) for a in items]
    with mp.get_context("spawn").Pool(procs) as pool:
        return pool.map(_one, items, chunksize=max(1, len(items) // (4 * procs)))
