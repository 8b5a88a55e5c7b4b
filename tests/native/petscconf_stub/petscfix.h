/* see petscconf.h in this directory: parse-only stub */
#ifndef PETSCFIX_STUB_H
#define PETSCFIX_STUB_H
#endif
