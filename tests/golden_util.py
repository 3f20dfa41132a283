import json
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def index():
    with open(os.path.join(GOLDEN, "golden.json")) as f:
        return json.load(f)


def load_image(entry):
    return np.load(os.path.join(GOLDEN, entry["file"]), allow_pickle=False)


def uniforms_bytes(entry):
    return np.array(entry["uniforms_u32"], dtype=np.uint32).tobytes()


def words(entry):
    return np.array(entry["words"], dtype=np.uint32)
