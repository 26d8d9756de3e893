registry = {}


def register(id, entry_point=None, kwargs=None, **_ignored):
    registry[id] = (entry_point, dict(kwargs or {}))
