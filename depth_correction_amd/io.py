"""Text files shared between processes: ``write`` / ``append`` serialised through an exclusive ``<path>.lock`` file.  The
experiment scripts collect result tables from parallel runs this way (scripts/model_poses_learning_icp:15; contract of the
reference's io.py:7-65: ``PathLock(path, interval, repeat)`` with ``lock`` / ``unlock`` / context-manager use, waiting a random
share of ``interval`` between attempts, ``repeat`` < 0 meaning for ever, ``PathLockException`` once the attempts are used up).

Written from that contract: the lock is taken with ``os.open(O_CREAT | O_EXCL)`` -- atomic on POSIX file systems -- and holds
the owner's pid, which makes a stale lock identifiable by a person looking at the directory."""
from __future__ import annotations

import itertools
import os
import random
import time

__all__ = ['write', 'append', 'PathLock', 'PathLockException']


class PathLockException(Exception):
    """The lock file stayed taken for all permitted attempts."""


class PathLock:
    lock_template = '%s.lock'

    def __init__(self, path, interval=1.0, repeat=-1):
        self.path = path
        self.lock_path = self.lock_template % path
        self.interval = interval
        self.repeat = repeat
        self.locked = False

    def _try_once(self):
        try:
            fd = os.open(self.lock_path, os.O_CREAT | os.O_EXCL | os.O_WRONLY, 0o644)
        except FileExistsError:
            return False
        with os.fdopen(fd, 'w') as f:
            f.write('%d\n' % os.getpid())
        return True

    def lock(self):
        assert not self.locked, 'lock() on a lock this object already holds'
        attempts = itertools.count() if self.repeat < 0 else range(self.repeat + 1)
        for _ in attempts:
            if self._try_once():
                self.locked = True
                return self
            time.sleep(self.interval * random.random())
        raise PathLockException('could not take %s' % self.lock_path)

    def unlock(self):
        assert self.locked and os.path.exists(self.lock_path)
        os.unlink(self.lock_path)
        self.locked = False

    def __enter__(self):
        return self.lock()

    def __exit__(self, *exc):
        if self.locked:
            self.unlock()
        return False


def _put(path, text, mode, create_dirs):
    folder = os.path.dirname(path)
    if create_dirs and folder:
        os.makedirs(folder, exist_ok=True)
    with PathLock(path), open(path, mode) as f:
        f.write(text)


def write(path, text, append=False, create_dirs=True):
    _put(path, text, 'a' if append else 'w', create_dirs)


def append(path, text, create_dirs=True):
    _put(path, text, 'a', create_dirs)
