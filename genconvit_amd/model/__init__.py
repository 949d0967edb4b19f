"""Host-side mirror of the reference's ``model`` package (same module and symbol names) backed by
the gfx950 HIP library.  ``import genconvit_amd.model as model`` / see INTEGRATION.md for the
drop-in aliasing used by the reference's ``prediction.py``."""
