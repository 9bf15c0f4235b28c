"""ORACLE -- restated rps.utilities (see ../__init__.py)."""
