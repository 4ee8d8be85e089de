"""Path and config resolution: ``paths_config.json``, dataset / model / inference directories.

Mirrors the reference's ``utils/data.py:13-132`` and ``utils/files.py:37-41`` so that the same
directory layout (``<dataset_path>/<dataset>/<subset>/{images,annotations,metadata}``,
``<dataset_path>/inference/<dataset>/<subset>/<model>/NNNN_results.pkl``,
``<model_path>/<type>/<name>/{config.json,calibration.json,...}``) keeps working.
"""
from __future__ import annotations

import glob
import json
import logging
import os
import sys
from typing import Dict, List


def find_existing_path(candidates: List[str]) -> str:
    for p in candidates:
        if os.path.exists(p):
            return p
    raise FileNotFoundError(candidates)


def load_paths_config() -> Dict:
    for base in ["."] + list(sys.path):
        f = os.path.join(base, "paths_config.json")
        if os.path.isfile(f):
            with open(f) as fh:
                return json.load(fh)
    raise FileNotFoundError("paths_config.json")


def get_dataset_base_path() -> str:
    return find_existing_path(load_paths_config()["dataset_path"])


def get_model_base_path() -> str:
    return find_existing_path(load_paths_config()["model_path"])


def get_inference_path(model_name: str, dataset: str, subset: str) -> str:
    return os.path.join(get_dataset_base_path(), "inference", dataset, subset, model_name)


def fetch_data_paths(dataset: str, subset: str) -> Dict[str, List[str]]:
    base = os.path.join(get_dataset_base_path(), dataset, subset)
    res = {k: sorted(glob.glob(os.path.join(base, d, pat)))
           for k, d, pat in (("images", "images", "*.png"), ("annotations", "annotations", "*.pkl"),
                             ("metadata", "metadata", "*.json"))}
    return res


def get_model_config_by_name(name: str):
    hits = glob.glob(os.path.join(get_model_base_path(), "*", name, "config.json"))
    if len(hits) > 1:
        logging.warning(f"found more than one model for {name}: {hits}")
    return hits[-1] if hits else None


def get_config_from_model_configs(name: str):
    for base in ["."] + list(sys.path):
        if os.path.isdir(os.path.join(base, "model_configs")):
            hits = glob.glob(os.path.join(base, "model_configs", "*", name)) + \
                glob.glob(os.path.join(base, "model_configs", "*", name + ".json"))
            return hits[-1] if hits else None
    return None


def resolve_model_config_path(config_file_or_model_name: str) -> str:
    """Full path -> a file in model_configs/*/ -> a stored model's config.json (reference ``utils/data.py:114-132``)."""
    if os.path.exists(config_file_or_model_name):
        return config_file_or_model_name
    f = get_config_from_model_configs(config_file_or_model_name) or get_model_config_by_name(config_file_or_model_name)
    if f is None:
        print(f"no model with name (or config with path) {config_file_or_model_name}")
        raise FileNotFoundError(config_file_or_model_name)
    return f


def make_if_not_exist(path, recursive: bool = False):
    for p in ([path] if isinstance(path, str) else path):
        os.makedirs(p, exist_ok=True) if recursive else (os.path.exists(p) or os.mkdir(p))
