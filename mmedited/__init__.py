"""Import-path shim: the reference's configs import their classes as `mmedited.models...`
(configs/001_*.py:6-8).  These modules re-export the MI355X implementations from `ciaosr_amd`."""
