"""emoasr_amd/hostenv.py: the CPU pool follows the container's quota"""
import os

import torch

from emoasr_amd import hostenv


def test_cpu_quota_is_read_and_respected(monkeypatch):
    q = hostenv.cpu_quota()
    assert q is None or q >= 1
    before = torch.get_num_threads()
    try:
        monkeypatch.setattr(hostenv, "_done", False)
        monkeypatch.setattr(hostenv, "cpu_quota", lambda: 16)
        monkeypatch.setenv("LOCAL_WORLD_SIZE", "1")
        torch.set_num_threads(max(before, 2))
        n = hostenv.respect_cpu_quota()
        assert 1 <= n <= 12
        assert hostenv.respect_cpu_quota() == n            # idempotent
        monkeypatch.setattr(hostenv, "_done", False)
        monkeypatch.setenv("LOCAL_WORLD_SIZE", "8")        # 8 ranks share 16 CPUs: 2 each -> one intra-op thread
        assert hostenv.respect_cpu_quota() == 1
        monkeypatch.setattr(hostenv, "_done", False)
        monkeypatch.setenv("EMOASR_CPU_THREADS", "3")
        assert hostenv.respect_cpu_quota() == 3
    finally:
        torch.set_num_threads(before)
