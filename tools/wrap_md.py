"""Re-wrap the paragraphs / list items of a Markdown file that contain a line longer than 120 characters (tables, headings
and code fences are left alone).  usage: python tools/wrap_md.py FILE..."""
import re
import sys
import textwrap

W = 120


def wrap_file(path):
    L = open(path, encoding="utf-8").read().split("\n")
    out, i, fence = [], 0, False
    item = re.compile(r"^(\s*)([-*]|\d+\.)\s+")
    while i < len(L):
        l = L[i]
        if l.lstrip().startswith("```"):
            fence = not fence
        if fence or not l.strip() or l.lstrip().startswith(("#", "|", "```", "<!--")):
            out.append(l)
            i += 1
            continue
        # a block: a list-item line or a paragraph line, plus its continuation lines
        m = item.match(l)
        first_indent = len(l) - len(l.lstrip())
        cont_indent = len(m.group(0)) if m else first_indent
        j = i
        while (j + 1 < len(L) and L[j + 1].strip() and not item.match(L[j + 1])
               and not L[j + 1].lstrip().startswith(("#", "|", "```", "<!--"))
               and (len(L[j + 1]) - len(L[j + 1].lstrip())) == cont_indent):
            j += 1
        block = L[i:j + 1]
        if any(len(x) > W for x in block):
            text = " ".join(x.strip() for x in block)
            head = l[:cont_indent] if m else " " * first_indent
            body = text[len(m.group(0).strip()) + 1:].lstrip() if m else text
            wrapped = textwrap.wrap(body, width=W - cont_indent, break_long_words=False, break_on_hyphens=False)
            out.append((l[:first_indent] + m.group(0).strip() + " " if m else head) + wrapped[0])
            out += [" " * cont_indent + w for w in wrapped[1:]]
        else:
            out += block
        i = j + 1
    open(path, "w", encoding="utf-8").write("\n".join(out))
    return [(k + 1, len(x)) for k, x in enumerate(out) if len(x) > W and not x.lstrip().startswith("|")]


if __name__ == "__main__":
    for p in sys.argv[1:]:
        print(p, wrap_file(p))
