"""Process-group setup, training-state checkpoints and the rank-0 logger, with the names and
behaviour of the reference's src/utils.py (setup_process :9-13, save/load_training_data :20-39,
Logger :42-79, TaskType :82-89).  The communication backend is RCCL (torch's "nccl" on ROCm)."""
import logging
import os
import sys

import torch
import torch.distributed as dist


def setup_process(rank, world_size, master_port="12355", backend=None):
    """One process per GPU on one node.  The reference hard-codes localhost + NCCL (utils.py:9-13);
    127.0.0.1 is used because container hostnames may not resolve, and `backend` can be overridden
    ("gloo") for CPU tests of the data-parallel logic."""
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ["MASTER_PORT"] = str(master_port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if backend is None:
        backend = "nccl" if torch.cuda.is_available() else "gloo"
    dist.init_process_group(backend, rank=rank, world_size=world_size)


def cleanup_process():
    if dist.is_initialized():
        dist.destroy_process_group()


def save_training_data(path, optimizer=None, scaler=None, epoch=None):
    state = {
        "optimizer": optimizer.state_dict() if optimizer is not None else None,
        "scaler": scaler.state_dict() if scaler is not None else None,
        "epoch": epoch,
    }
    torch.save(state, os.path.join(path, "training_data.pt"))


def load_training_data(path, optimizer=None, scaler=None, map_location=None):
    state = torch.load(os.path.join(path, "training_data.pt"), map_location=map_location)
    if optimizer is not None and state.get("optimizer") is not None:
        optimizer.load_state_dict(state["optimizer"])
    if scaler is not None and state.get("scaler") is not None:
        scaler.load_state_dict(state["scaler"])
    return state


class Logger:
    """stdout (+ optional file) logger that is silent when disabled (non-zero ranks)."""

    def __init__(self, log_dir=None, enabled=True, pad_length=50):
        self._pad_length = pad_length
        self._logger = None
        if enabled:
            lg = logging.getLogger("kmbart")
            lg.setLevel(logging.DEBUG)
            lg.propagate = False
            if not lg.handlers:
                sh = logging.StreamHandler(sys.stdout)
                sh.flush = sys.stdout.flush
                lg.addHandler(sh)
            if log_dir is not None:
                fh = logging.FileHandler(log_dir)
                fh.setFormatter(logging.Formatter("%(asctime)s %(levelname)s %(message)s"))
                lg.addHandler(fh)
            self._logger = lg

    def info(self, message, pad=False):
        if self._logger is None:
            return
        if pad:
            message = (" " + message + " ").center(self._pad_length, "=")
        self._logger.info(message)

    def line(self):
        if self._logger is not None:
            self._logger.info("=" * self._pad_length)


class TaskType:
    AFTER = "after"
    BEFORE = "before"
    INTENT = "intent"
    CAPTION = "caption"
    REGION_CAPTION = "region_caption"
    ALL_TYPES = {AFTER, BEFORE, INTENT, CAPTION, REGION_CAPTION}
