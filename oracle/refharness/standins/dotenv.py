"""Stand-in for python-dotenv."""


def load_dotenv(*_a, **_kw):
    return False
