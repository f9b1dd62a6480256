"""rigl.rigl_tf2 -> the TF2-style front-end of rigl_amd (HIP-backed)."""
