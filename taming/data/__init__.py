"""Import-path alias package: host-side data helpers the reference's scripts import from `taming.data`."""
