"""Mirror of the reference's SOTAS/ tree (only the modules on the accelerated path)."""
