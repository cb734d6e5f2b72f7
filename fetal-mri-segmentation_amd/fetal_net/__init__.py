"""host-side mirror of the reference fetal_net API (hot path only)"""
