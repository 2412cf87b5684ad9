"""CSV logger with the reference's `log(dict, step)` surface (pipeline/utils/logger.py:144-165)."""
import csv
import os


class Logger:
    def __init__(self, config=None):
        self.config = config

    def log(self, update_dict, step, split=""):
        return {(f"{split}/{k}" if split else k): (float(v) if hasattr(v, "__float__") else v) for k, v in update_dict.items()}


class CSVLogger(Logger):
    def __init__(self, save_dir, fname="metrics", config=None, **kwargs):
        super().__init__(config)
        self.save_dir, self.fname, self.rows = save_dir, fname, []

    def log(self, update_dict, step, split=""):
        row = super().log(update_dict, step, split)
        row["step"] = step
        self.rows.append(row)
        keys = sorted({k for r in self.rows for k in r})
        os.makedirs(self.save_dir, exist_ok=True)
        with open(os.path.join(self.save_dir, f"{self.fname}.csv"), "w", newline="") as f:  # rewritten each step, like the reference
            w = csv.DictWriter(f, fieldnames=keys)
            w.writeheader()
            w.writerows(self.rows)
