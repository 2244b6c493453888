"""Text files shared between processes (io.py:7-65): ``write`` / ``append`` under an exclusive ``<path>.lock`` file, which
the experiment scripts use to collect result tables from parallel runs (scripts/model_poses_learning_icp:15)."""
from __future__ import annotations

import os
import random
import time

__all__ = ['write', 'append', 'PathLock', 'PathLockException']


class PathLockException(Exception):
    pass


class PathLock(object):
    """Context manager holding ``<path>.lock`` (created with mode 'x'); waits a random fraction of ``interval`` between
    attempts, ``repeat`` < 0: for ever."""

    lock_template = '%s.lock'

    def __init__(self, path, interval=1.0, repeat=-1):
        self.path, self.lock_path = path, PathLock.lock_template % path
        self.locked, self.interval, self.repeat = False, interval, repeat

    def lock(self):
        assert not self.locked
        attempt = 0
        while self.repeat < 0 or attempt <= self.repeat:
            attempt += 1
            try:
                with open(self.lock_path, 'x'):
                    pass
                self.locked = True
                return self
            except FileExistsError:
                time.sleep(random.random() * self.interval)
        raise PathLockException()

    def unlock(self):
        assert self.locked and os.path.exists(self.lock_path)
        os.remove(self.lock_path)
        self.locked = False

    def __enter__(self):
        return self.lock()

    def __exit__(self, exc_type, exc_val, exc_tb):
        if self.locked:
            self.unlock()


def write(path, text, append=False, create_dirs=True):
    if create_dirs:
        os.makedirs(os.path.dirname(path), exist_ok=True)
    with PathLock(path):
        with open(path, 'a' if append else 'w') as f:
            f.write(text)


def append(path, text, create_dirs=True):
    write(path, text, append=True, create_dirs=create_dirs)
