"""Import-path alias package: the reference's config `target:` strings resolve to frido_amd classes."""
