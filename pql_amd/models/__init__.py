"""Class-name -> file registry, same contract as pql/models/__init__.py:5-6."""
from pathlib import Path

from pql_amd.utils.common import list_class_names

cur_path = Path(__file__).resolve().parent
model_name_to_path = list_class_names(cur_path)
