"""The host library binding lives in the package (afec_amd/hostlib.py); tests import it from here."""
from afec_amd.hostlib import crawl, lib, wave_probe, wave_probe_file  # noqa: F401
