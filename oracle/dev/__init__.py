"""Build-container-only scripts that import /root/reference to generate the golden fixtures under tests/golden/."""
