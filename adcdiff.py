"""Import alias: ``import adcdiff`` loads the package directory whose name (the upstream repository's
name plus ``_amd``) is not a valid Python identifier."""
import importlib.util
import os
import sys

_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)),
                    "2025-iflytek-adc-generation-challenge-based-on-diffusion-models-and-reinforcement-learning_amd")
_spec = importlib.util.spec_from_file_location("adcdiff", os.path.join(_DIR, "__init__.py"),
                                               submodule_search_locations=[_DIR])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["adcdiff"] = _mod
_spec.loader.exec_module(_mod)
