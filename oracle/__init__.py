"""Test-only CPU oracle (see pql_ref_cpu.py). Never imported by pql_amd/."""
