"""Configuration loader — mirrors the reference's ``model/config.py:6-10``.

The reference opens ``model/config.yaml`` relative to the current working directory; that is kept
(so an existing checkout keeps reading its own file) with a fallback to the copy shipped next to
this module when the cwd-relative file does not exist.
"""
import os

import yaml

_HERE = os.path.dirname(os.path.abspath(__file__))


def load_config():
    path = os.path.join("model", "config.yaml")
    if not os.path.exists(path):
        path = os.path.join(_HERE, "config.yaml")
    with open(path) as file:
        config = yaml.safe_load(file)
    return config
