"""Helpers shared by oracle and GPU parity tests."""


def simplify(node):
    """Strip op ids from the canonical materialization: map -> {key: [values]}, list -> {"list": [[values]]}."""
    if isinstance(node, dict):  # a map object: key -> list of visible values
        return {k: [simplify_value(v) for v in vs] for k, vs in node.items()}
    if isinstance(node, list):
        return [[simplify_value(v) for v in vs] for vs in node]
    return node


def simplify_value(v):
    for tag in ("str", "int", "uint", "bool", "null"):
        if tag in v:
            return v[tag]
    if "counter" in v:
        return {"counter": v["counter"]}
    if "map" in v:
        return {"map": simplify(v["map"])}
    if "table" in v:
        return {"table": simplify(v["table"])}
    if "list" in v:
        return {"list": simplify(v["list"])}
    if "text" in v:
        return {"text": v["text"]}
    return {k: val for k, val in v.items() if k != "id"}
