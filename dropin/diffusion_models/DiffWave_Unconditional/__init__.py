"""Drop-in package: reference import paths -> audiopure_amd."""
