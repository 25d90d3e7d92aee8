"""Class-name -> file registry for algorithms, same contract as pql/algo/__init__.py."""
from pathlib import Path

from pql_amd.utils.common import list_class_names

cur_path = Path(__file__).resolve().parent
alg_name_to_path = list_class_names(cur_path)
