"""Alias package: the reference's import paths for the hot-path entry points, re-exported from
``arah_release_amd``.  Nothing is implemented here."""
