"""teamoflow_amd - MI355X-native engine behind TeAMOFlow's ``teamoflow.mf.MatrixFactorization``.

``from teamoflow_amd.mf import matrix_factorization`` mirrors ``from teamoflow.mf import ...`` of the
reference (src/teamoflow/mf/__init__.py:1-9); the top-level ``teamoflow`` package in this repo is a
thin alias so reference-style imports keep working.
"""
__version__ = '0.1.0'
